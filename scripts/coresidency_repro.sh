#!/bin/bash
# runs scripts/coresidency_repro over co-runner kinds, victims and CU-mask / XNACK / same-process settings; one JSON line each
# usage (GPU box, repo root): bash scripts/coresidency_repro.sh > profiles/r04_coresidency_repro.jsonl
set -u
B=./scripts/coresidency_repro
[ -x $B ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $B scripts/coresidency_repro.hip -lpthread
N=${N:-600}
( /opt/rocm/bin/rocm-smi --showdriverversion --showfwinfo 2>/dev/null | grep -E "Driver|MEC|RLC firmware|SDMA firmware|SMC|SOS" | head -8 | sed 's/^/# /' ) || true
echo "# rocm $(cat /opt/rocm/.info/version 2>/dev/null), kernel $(uname -r)"
short() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
keep={k:d[k] for k in ('corunner','corunner_in','launches','victim_cu_mask','corunner_cu_mask','xnack_env','msda_bad_launches','msda_bad_words','gather_bad_launches','regs_bad_launches','trans_bad_launches','gcn_arch','hip_runtime')}
keep['msda_bad_lanes']=[i for i,v in enumerate(d['msda_bad_lanes_histogram']) if v]
print(json.dumps(keep))"; }
# the deformable-attention gather of the product (victim) next to register-only co-runners in ANOTHER process
for k in none bf16_16x16x32 f16_16x16x32 bf16_32x32x16 f16_32x32x16 f32_32x32x2 valu; do timeout 200 $B $k $N msda | short; done
# clean victims (plain gather, idle registers, transcendental / integer / packed-math chains) next to the one co-runner that bites
for v in both trans imul misc; do timeout 200 $B bf16_16x16x32 $N $v | short; done
# disjoint CU ranges: the corruption must vanish
REPRO_VICTIM_CU_MASK=0:0-127 REPRO_SPIN_CU_MASK=0:128-255 timeout 200 $B bf16_16x16x32 $N msda | short
# XNACK on / off
HSA_XNACK=1 timeout 200 $B bf16_16x16x32 $N msda | short
HSA_XNACK=0 timeout 200 $B bf16_16x16x32 $N msda | short
# the co-runner inside the SAME process on a second stream
REPRO_SAME_PROCESS=1 timeout 200 $B bf16_16x16x32 $N msda | short
REPRO_SAME_PROCESS=1 timeout 200 $B bf16_32x32x16 $N msda | short
# round 4: the instruction of the f16x2 split kernels, other process and second stream of the same process
REPRO_SAME_PROCESS=1 timeout 200 $B f16_16x16x32 $N msda | short
REPRO_VICTIM_CU_MASK=0:0-127 REPRO_SPIN_CU_MASK=0:128-255 timeout 200 $B f16_16x16x32 $N msda | short
