#!/bin/bash
# pmc_sq.sh: where the waves of the dominant kernels spend their cycles - one rocprofv3 --pmc pass (8 SQ slots) over one train step
# of bench.py (program directly behind "--"; counters in their own run, no other trace domain).  SQ_WAVE_CYCLES / SQ_WAIT_* /
# SQ_ACTIVE_INST_* count quad-cycles per wave, SQ_VALU_MFMA_BUSY_CYCLES cycles per SIMD (MI355X_MICROARCH.md).
# usage (GPU box, repo root): bash profiles/pmc_sq.sh > gpurun_out/r04_pmc_sq.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_sq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-per-pass > $O/run.log 2>&1
f=$(find $O -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections, os
sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'profiles'))
from parse_pmc import short
d = collections.defaultdict(lambda: collections.Counter())
n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r['Kernel_Name'])
    d[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r.get('Dispatch_Id'), k)
    if key not in seen:
        seen.add(key)
        n[k] += 1
rows = sorted(d.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES'])[:16]
print('%-34s %5s %9s %7s %9s %8s %8s %9s %9s %9s' % ('kernel (one step)', 'n', 'wave Gcyc', 'wait', 'wait_inst', 'active', 'wait_lds', 'lds_confl', 'lds_active', 'mfma_busy'))
for k, c in rows:
    w = c['SQ_WAVE_CYCLES'] or 1.0
    print('%-34s %5d %9.2f %7.2f %9.2f %8.2f %8.2f %9.3f %9.3f %9.3f' % (
        k[:34], n[k], w / 1e9, c['SQ_WAIT_ANY'] / w, c['SQ_WAIT_INST_ANY'] / w, c['SQ_ACTIVE_INST_ANY'] / w, c['SQ_WAIT_INST_LDS'] / w,
        c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1.0), c['SQ_LDS_IDX_ACTIVE'] / w, c['SQ_VALU_MFMA_BUSY_CYCLES'] / w))
print('wait / wait_inst / active / wait_lds: fractions of SQ_WAVE_CYCLES; lds_confl = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;')
print('lds_active, mfma_busy = SQ_LDS_IDX_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES per wave quad-cycle (compare across kernels, not to 1)')
PY
rm -rf $O
