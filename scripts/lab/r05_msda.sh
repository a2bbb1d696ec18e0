cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_msda; mkdir -p $O
{
for v in off on map off on map; do
  echo "== frames 32 PVSG_MSDA_LDS=$v"
  PVSG_MSDA_LDS=$v python scripts/kbench.py msda --frames 32 2>/dev/null | grep msda_fused
done
} 2>&1 | tee $O/msda_lds_ab2.txt
