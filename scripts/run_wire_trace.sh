#!/bin/bash
TILES=${TILES:-512} VARIANTS="${VARIANTS:-}" bash "$(dirname "$0")/ubench/wire_trace.sh" run
