#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/nf_prof.sh -> per-kernel averages of the NoiseFlow fit step
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/nfp -o nf -- python3 /root/repo/tools/nf_fit_bench.py --steps 20 > /dev/null 2>&1
python3 - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/nfp/**/nf_results.db', recursive=True)[0])
rows = list(db.execute("select name, total_calls, total_duration, average from top_kernels"))
tot = sum(r[2] for r in rows)
print('total kernel time (us):', tot / 1e3)
for r in rows[:14]:
    print('%-70s calls %6d avg %8.2f us  %5.1f%%' % (r[0][:70], r[1], r[3] / 1e3 if r[3] > 1e3 else r[3], 100 * r[2] / tot))
PY
