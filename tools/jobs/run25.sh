set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r02/gputest_j.log 2>&1 || (tail -40 gpurun_out/r02/gputest_j.log; exit 1)
tail -10 gpurun_out/r02/gputest_j.log
python -c "import __graft_entry__ as g; g.smoke()"
echo done
