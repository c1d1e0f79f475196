# scratch job: edit for the experiment at hand (gpurun -- 'bash tools/jobs/quick.sh')
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()"
