# processes that end right after creating a context (the warm-up thread is still busy): must exit cleanly
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
for i in 1 2 3; do
python3 -c "
import sys; sys.path.insert(0, '.')
from ipde_amd.device import get_context
get_context(); print('context made, exiting')"
done
python3 -c "
import sys; sys.path.insert(0, '.')
import torch
from ipde_amd.device import get_context
c = get_context(); torch.ones(4, device='cuda').sum().item(); raise SystemExit(0)"
echo exits ok
