#!/bin/bash
# wave-cycle accounting of the split GEMM kernels, both forms (PMC passes, kernel-trace only); run on the GPU box from the repo root
#   scripts/lab/pmc_split.sh [gemm|conv]  ->  gpurun_out/pmc_split_<what>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WHAT=${1:-gemm}
CMD="python $R/scripts/split_ab.py $WHAT 2"
run() { d=$1; shift; rm -rf /tmp/$d; timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE "$@" --kernel-trace --output-format csv -d /tmp/$d -- $CMD > /dev/null 2>&1 || echo "pass $d failed / timed out"; }
if [ "${2:-sq}" = "mem" ]; then          # memory side: texture addresser, vector L1, L2 (<= 4 counters of a block per pass)
run ps1 TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run ps2 TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum
run ps3 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run ps4 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum
run ps5 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
else
run ps1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
run ps2 SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM
run ps3 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS
run ps4 SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
run ps5 SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAVES
fi
cd $R
python - "$WHAT" <<'PY' | tee gpurun_out/pmc_split_${WHAT}_${2:-sq}.txt
import collections, csv, glob, re, sys
for d in ('/tmp/ps1', '/tmp/ps2', '/tmp/ps3', '/tmp/ps4', '/tmp/ps5'):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    t = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    if not f:
        print(d, 'no counter file'); continue
    dur = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(t[0]))} if t else {}
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        name = re.sub(r'^void |pvsg::|\(anonymous namespace\)::', '', r['Kernel_Name']); name = re.sub(r'\(.*', '', name)
        if 'k32_kernel' not in name: continue
        c = agg.setdefault((name, r['Grid_Size']), collections.defaultdict(list))
        c[r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': c['dur_us'].append(dur.get(r['Dispatch_Id'], float('nan')))
    for (name, grid), c in agg.items():
        med = {k: sorted(v)[len(v) // 2] for k, v in c.items()}
        extra = ''
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in med:
            cyc = med['GRBM_GUI_ACTIVE'] / 8.0
            extra = ' clk_GHz=%.2f mfma_busy=%.1f%% wait_any=%.1f%% wait_inst=%.1f%% active=%.1f%%' % (
                cyc / med['dur_us'] / 1e3, 100 * med['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc), 100 * med['SQ_WAIT_ANY'] / med['SQ_WAVE_CYCLES'],
                100 * med['SQ_WAIT_INST_ANY'] / med['SQ_WAVE_CYCLES'], 100 * med['SQ_ACTIVE_INST_ANY'] / med['SQ_WAVE_CYCLES'])
        print('%-48s %9s n=%d ' % (name[:48], grid, len(c['GRBM_GUI_ACTIVE'])) + '  '.join('%s=%.5g' % (k, v) for k, v in med.items()) + extra)
PY
