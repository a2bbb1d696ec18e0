#!/bin/bash
# memory-side counters of the GEMM lab kernels (two PMC passes); run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TA|TCC|TD)_[A-Z0-9_]+" | sort -u > $R/gpurun_out/r3h/counters_mem.txt
rocprofv3 --pmc TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_m1 -- $R/scripts/lab/gemm_lab 1 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_m2 -- $R/scripts/lab/gemm_lab 1 > /dev/null 2>&1
cd $R
python - <<'PY'
import collections, csv, glob, re
for d in ('/tmp/pmc_m1', '/tmp/pmc_m2'):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f:
        print(d, 'no counter file'); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        name = re.sub(r'\(.*', '', r['Kernel_Name']); name = re.sub(r'^void |pvsg::|\(anonymous namespace\)::', '', name)
        if 'gemm' not in name: continue
        agg.setdefault((name, r['Grid_Size']), collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
    for (name, grid), c in agg.items():
        print('%-40s %10s ' % (name[:40], grid) + '  '.join('%s=%.4g' % (k, sum(v) / len(v)) for k, v in c.items()))
PY
