#!/bin/bash
# Build container only: copy the five Python files of the reference's MSDeformAttn package into .ref_stage/ (git-ignored, NOT
# gpurun-ignored) so that tools/reference_on_hip.py can run them, unmodified, on the GPU box, where /root/reference does not exist.
# Delete .ref_stage afterwards; it is never committed.
set -e
src=/root/reference/projects/UNINEXT/uninext/models/deformable_detr/ops
dst=/root/repo/.ref_stage/ops
rm -rf /root/repo/.ref_stage
mkdir -p $dst/functions $dst/modules
cp $src/test.py $dst/
cp $src/functions/__init__.py $src/functions/ms_deform_attn_func.py $dst/functions/
cp $src/modules/__init__.py $src/modules/ms_deform_attn.py $dst/modules/
echo staged: $(find $dst -name "*.py" | wc -l) files
