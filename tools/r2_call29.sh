#!/bin/bash
# sanity after reverting the 6-wave key-split experiment: default bench lines
set -u
cd $GRAFT_REPO_ROOT
for b in "1 16 5 2" "2 16 3 1"; do set -- $b
timeout 300 python bench.py --steps $3 --warmup $4 --no-cpu-baseline --batch $1 --nfe $2 2>/tmp/err | python -c "
import json,sys
t=sys.stdin.read().strip()
print('B=$1', json.loads(t.splitlines()[-1])['ms_per_step'] if t else 'NO OUTPUT')"; tail -2 /tmp/err | cut -c1-200
done
