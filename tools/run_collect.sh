#!/bin/bash
# Local wrapper (this container): labels the profiles with the commit they were collected at.  Refuses to run
# when the product sources differ from HEAD, so that profiles/counters.json never describes unreleased code.
#   bash tools/run_collect.sh r02
set -e
tag=${1:-r02}
cd "$(dirname "$0")/.."
if ! git diff --quiet HEAD -- peapods_amd include bench.py oracle tools/collect_profiles.sh tools/summarize_profiles.py; then
    echo "working tree differs from HEAD in the measured sources: commit first" >&2
    exit 1
fi
git rev-parse --short=12 HEAD > profiles/collect_head.txt
/usr/local/graft/bin/gpurun --timeout 1150 -- "bash tools/collect_profiles.sh $tag"
src=gpurun_out/prof_$tag
if [ "$(cat profiles/collect_head.txt)" != "$(git rev-parse --short=12 HEAD)" ]; then
    echo "HEAD moved while collecting" >&2
    exit 1
fi
python3 tools/summarize_profiles.py "$src" "$tag"
