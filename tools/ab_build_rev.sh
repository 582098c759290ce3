#!/bin/bash
# build the library of a git revision (default HEAD) as tools/_build/libecg_hip_<name>.so for same-box A/B runs
REV=${1:-HEAD}; NAME=${2:-prev}
set -e
R=$(git rev-parse --show-toplevel)
rm -rf /tmp/ecg_prev && git -C "$R" worktree prune && git -C "$R" worktree add -f /tmp/ecg_prev "$REV" > /dev/null 2>&1
make -s -j8 -C /tmp/ecg_prev/ptbxl-multimodal_amd/csrc > /dev/null
mkdir -p "$R/tools/_build" && cp /tmp/ecg_prev/ptbxl-multimodal_amd/lib/libecg_hip.so "$R/tools/_build/libecg_hip_$NAME.so"
git -C "$R" worktree remove --force /tmp/ecg_prev
echo "built $NAME from $REV"
