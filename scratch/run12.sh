cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/la
timeout 30 ./scratch/potf2_stamps > gpurun_out/la/lock_new.txt 2>&1
timeout 30 ./scratch/potf2_stamps_old > gpurun_out/la/lock_old.txt 2>&1
timeout 30 ./scratch/potf2_la_events > gpurun_out/la/lock_ev.txt 2>&1
echo "--- hand-scheduled"; sed -n 1,11p gpurun_out/la/lock_new.txt; grep "^info\|potf2_trsm m\|potf2x2 rep 2" gpurun_out/la/lock_new.txt
echo "--- compiler-scheduled"; sed -n 1,11p gpurun_out/la/lock_old.txt; grep "^info\|potf2_trsm m\|potf2x2 rep 2" gpurun_out/la/lock_old.txt
echo "--- events, no stamps"; grep -v "^  " gpurun_out/la/lock_ev.txt
