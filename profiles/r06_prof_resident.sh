cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for r in 1 0; do
  GPRN_BATCH_TIMERS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_resident_$r -o t -- python3 $R/profiles/batch_run.py 512 3 2 128 3 $r > $R/gpurun_out/r06_resident_${r}_prof.log 2>&1
  tail -2 $R/gpurun_out/r06_resident_${r}_prof.log
done
