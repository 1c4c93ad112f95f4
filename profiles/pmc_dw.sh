R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for v in base dw2 dw64; do
  O=$R/gpurun_out/pmc_dw_$v; mkdir -p $O
  if [ $v = base ]; then unset VKAS_LIB_PATH; else export VKAS_LIB_PATH=$R/build_variants/libvkas_$v.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --output-format csv -d $O -o p -- python3 $R/profiles/bench_dw.py --quick > $O/run.log 2>&1
  f=$(find $O -name "*counter_collection.csv" | head -1)
  python3 - "$f" $v <<'PY'
import csv, sys, collections
d = collections.Counter(); n = 0
for r in csv.DictReader(open(sys.argv[1])):
    if 'dwconv7x7_mfma_kernel' not in r['Kernel_Name']: continue
    d[r['Counter_Name']] += float(r['Counter_Value'])
w = d['SQ_WAVE_CYCLES'] or 1
print(sys.argv[2], 'wave Gcyc %.3f wait %.2f wait_inst %.2f active %.2f wait_lds %.2f lds_conflict/lds_active %.3f lds_active/wave %.3f addr_conflict %.0f' % (
    w / 1e9, d['SQ_WAIT_ANY'] / w, d['SQ_WAIT_INST_ANY'] / w, d['SQ_ACTIVE_INST_ANY'] / w, d['SQ_WAIT_INST_LDS'] / w,
    d['SQ_LDS_BANK_CONFLICT'] / max(d['SQ_LDS_IDX_ACTIVE'], 1), d['SQ_LDS_IDX_ACTIVE'] / w, d['SQ_LDS_ADDR_CONFLICT']))
PY
  rm -rf $O
done
