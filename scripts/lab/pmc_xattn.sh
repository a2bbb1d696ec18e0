#!/bin/bash
# wave-cycle accounting of the attention kernel (PMC passes, kernel-trace only); run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/xpmc
rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z0-9_]+" | sort -u > $R/gpurun_out/xpmc/sq_counters.txt
export KBENCH_XATTN_LEVEL=2
CMD="python $R/scripts/kbench.py xattn --frames 32"
run() { d=$1; shift; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES "$@" --kernel-trace --output-format csv -d /tmp/$d -- $CMD > /dev/null 2>&1; }
run px1 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
run px2 SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM
run px3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS
run px4 SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
run px5 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAVES
cd $R
python - <<'PY'
import collections, csv, glob, re
for d in ('/tmp/px1', '/tmp/px2', '/tmp/px3', '/tmp/px4', '/tmp/px5'):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f:
        print(d, 'no counter file'); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        name = re.sub(r'\(.*', '', r['Kernel_Name']); name = re.sub(r'^void |pvsg::|\(anonymous namespace\)::', '', name)
        if 'xattn_partial' not in name: continue
        agg.setdefault((name, r['Grid_Size']), collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
    for (name, grid), c in agg.items():
        print('%-40s %10s n=%d ' % (name[:40], grid, len(list(c.values())[0])) + '  '.join('%s=%.5g' % (k, sorted(v)[len(v)//2]) for k, v in c.items()))
PY
