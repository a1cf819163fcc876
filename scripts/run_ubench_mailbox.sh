#!/bin/bash
# runs the prebuilt mailbox micro-benchmark (built here: hipcc --offload-arch=gfx950 -O3 -w -o ubench_mailbox.bin ubench_mailbox.hip)
cd "$(dirname "$0")/ubench"
timeout 120 ./ubench_mailbox.bin
