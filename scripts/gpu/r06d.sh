cd $GRAFT_REPO_ROOT; O=gpurun_out/r06d; mkdir -p $O
export BAZ_MUSIC_LAB_LIB=quick
for rep in 1 2; do
for v in 1 14 15 1 14 15; do
  BAZ_MUSIC_SCAN_VARIANT=$v python tests/lab/scan_ablate.py one 2>/dev/null | tail -1
done; done | tee $O/prio.txt
