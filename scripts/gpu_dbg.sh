#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for c in 2,64,320,3,1,192 2,64,320,3,1,192 2,32,320,3,1,360 3,270,320,3,2,360 2,320,640,3,16,343 2,48,320,1,1,192; do
  echo "== $c"; timeout 120 python scripts/dbg_wide.py $c 2>&1 | grep -E "rel_l2|fault|Error|error" | head -3
done
