#!/bin/bash
# Run ON THE GPU BOX: the profile passes of a round for the headline line and the two sibling coders whose kernels share
# the range decoders' machinery (tools/profile_round.sh for each).  `bash tools/profile_all.sh r03b`
set -e -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
bash "$ROOT/tools/profile_round.sh" "$TAG"
bash "$ROOT/tools/profile_round.sh" "${TAG}_rans1" --coder rans --workload zipf
bash "$ROOT/tools/profile_round.sh" "${TAG}_static" --coder static --workload zipf
