#!/bin/bash
# round 5, call 10: full -m gpu suite + smoke on the final sources, then the evidence pass
mkdir -p gpurun_out/c10
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 200 > gpurun_out/c10/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/c10/pytest.log
grep -v "^\s*$" gpurun_out/c10/pytest.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c10/smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/c10/smoke.log; tail -3 gpurun_out/c10/smoke.log
bash tools/evidence_pass.sh > gpurun_out/c10/evidence.log 2>&1
tail -25 gpurun_out/c10/evidence.log
