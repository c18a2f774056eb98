#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <outdir> "<counters>" -- <program and args>
# One rocprofv3 --pmc pass (no other trace domains), CSV output.
out=$1; ctr=$2; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out" -- "$@"
