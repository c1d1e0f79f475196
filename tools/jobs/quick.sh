# scratch: whatever the current experiment needs
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
timeout -k 10 900 python3 tools/paper_table.py > $O/paper_table.jsonl 2> $O/paper_table.err || { tail -20 $O/paper_table.err; exit 1; }
cut -c1-400 $O/paper_table.jsonl
