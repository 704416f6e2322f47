#!/bin/bash
# round 3, GPU call H2: stall probe
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for n in 1 3; do timeout 300 python tools/stall_probe.py --inflight $n --steps 9000 2>&1 | tail -2 | cut -c1-1200; done
timeout 300 python tools/stall_probe.py --inflight 3 --steps 6000 --kind train 2>&1 | tail -2 | cut -c1-1200
timeout 300 python tools/stall_probe.py --inflight 6 --steps 6000 --kind train --render-res 1024 2>&1 | tail -2 | cut -c1-1200
