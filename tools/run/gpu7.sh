export TMPDIR=/tmp
VIT_COMMIT=$(cat .commit_id 2>/dev/null) bash tools/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1
echo "collect rc=$?"; tail -20 gpurun_out/r04_collect.log; ls gpurun_out/profiles/r04 2>/dev/null
