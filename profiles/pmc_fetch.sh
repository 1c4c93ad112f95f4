#!/bin/bash
# pmc_fetch.sh <label> [VKAS_LIB_PATH]: FETCH_SIZE per launch of the slab kernels in profiles/bench_slab.py for one library build
# (rocprofv3 --pmc in its own run, program directly behind "--"; bytes = 2 x KiB x 1024 on gfx950, MI355X_MICROARCH.md)
label=$1
export VKAS_LIB_PATH=${2:-}
[ -z "$VKAS_LIB_PATH" ] && unset VKAS_LIB_PATH
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_$label; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -o p -- python3 $R/profiles/bench_slab.py > $O/run.log 2>&1
f=$(find $O -name "*counter_collection.csv" | head -1)
python3 - "$f" "$label" <<'PY'
import csv, sys, collections, re
d = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] != 'FETCH_SIZE':
        continue
    m = re.search(r'(conv3x3_slab_mfma_kernel|conv3x3_wgrad_slab_kernel)I(Li\d+E(?:Lb\d+E)?)', r['Kernel_Name'])
    if not m:
        continue
    k = m.group(1) + '<' + ','.join(re.findall(r'\d+', m.group(2))) + '>' + ' grid ' + r.get('Grid_Size', '?')
    d[k][0] += 1
    d[k][1] += float(r['Counter_Value'])
for k, (n, v) in sorted(d.items()):
    print('%-12s %-50s launches %3d  fetch %.3f GB per launch' % (sys.argv[2], k, n, 2.0 * v * 1024.0 / n / 1e9))
PY
