#!/bin/bash
TILES=${TILES:-512} VARIANTS="${VARIANTS:-EARLY}" bash "$(dirname "$0")/ubench/wire_trace.sh" run
