#!/bin/bash
# GPU box: 12 processes of the plain streaming kernel, 6 of the classification timing -- is the spread per process a platform property?
cd "$GRAFT_REPO_ROOT"
for i in $(seq 12); do tools/micro/stream154; done
for i in $(seq 6); do timeout 120 python tools/classify_time.py 2>&1 | tail -1; done
