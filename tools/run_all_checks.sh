#!/bin/bash
# Everything a reviewer would run, in order.  On a machine without a GPU only the first three
# steps do anything; on the MI355X box (e.g. through `gpurun -- 'bash tools/run_all_checks.sh'`)
# the GPU steps follow.
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
python -m pytest tests -x -q -m "not gpu"
if python -c "import torch, sys; sys.exit(0 if torch.cuda.is_available() else 1)"; then
  python -m pytest tests -x -q -m gpu
  python -c "import __graft_entry__ as g; g.smoke()"
  python bench.py
else
  echo "no HIP device here: GPU tests, smoke() and bench.py skipped"
fi
