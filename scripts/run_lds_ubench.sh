#!/bin/bash
# scripts/ubench/ubench_lds.hip on the GPU box (built here by hipcc into scripts/ubench/variants/): cycles per LDS wave-instruction
cd "$(dirname "$0")/.."
timeout 120 scripts/ubench/variants/ubench_lds
