set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputest6.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -15 gpurun_out/r04/gputest6.log
