cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/la
timeout 30 ./scratch/chain_probe | head -2
GPT_POTF2_LA=1 timeout 30 ./scratch/potf2_la_stamps > gpurun_out/la/la.txt 2>&1; echo "rc $?" >> gpurun_out/la/la.txt
GPT_POTF2_LA=1 timeout 30 ./scratch/potf2_la_events > gpurun_out/la/la_ev.txt 2>&1; echo "rc $?" >> gpurun_out/la/la_ev.txt
GPT_POTF2_LA=0 timeout 30 ./scratch/potf2_la_events > gpurun_out/la/lock.txt 2>&1; echo "rc $?" >> gpurun_out/la/lock.txt
grep -A10 "^rep 3" gpurun_out/la/la.txt; echo "--- no stamps:"; grep -v "^  " gpurun_out/la/la_ev.txt;  echo "--- lock-step:"; grep -v "^  " gpurun_out/la/lock.txt | head -5
