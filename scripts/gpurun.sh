#!/bin/bash
# build the library HERE (hipcc cross-compiles; a stale libmaed_hip.so would travel otherwise), then run a script on the GPU box
# usage: scripts/gpurun.sh <timeout seconds> <command...>
set -e
cd "$(dirname "$0")/.."
python -m maed_amd.build 2>&1 | tail -1
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
