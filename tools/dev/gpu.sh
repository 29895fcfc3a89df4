#!/bin/bash
# Dev tool: rebuild everything that travels to the GPU box (gfx950 library, host module, oracle), then gpurun the command.
#   tools/dev/gpu.sh <timeout_s> '<command>'
set -e
cd "$(dirname "$0")/../.."
python -c "import __graft_entry__ as g; g.build()"
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
