#!/bin/bash
# Development loop on the GPU box: rebuild kernel files with extra compiler flags, relink, run a command.
#   profiles/dev_variant.sh pass_royale_bloom_quad="-DRC_BQ_WAVES=8" pass_royale_bloom="-DRC_BQ_COST_TWO=26" -- python profiles/dev_pass_check.py
# Files named without flags are rebuilt plain (to drop a previous variant's object).
set -e
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ARGS=()
while [ "$1" != "--" ]; do
  F="${1%%=*}"; D=""; [[ "$1" == *=* ]] && D="${1#*=}"
  touch "retrocapture_amd/csrc/kernels/$F.hip"
  ARGS+=("EXTRA_$F=$D")
  shift
done
shift
make -s -C retrocapture_amd/csrc -j16 "${ARGS[@]}" > /dev/null
"$@"
