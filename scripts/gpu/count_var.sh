#!/bin/bash
# counting kernel A/B builds (gsn_amd/lib/variants/libgsn_hip_c*.so against the product library; GSN_COUNT_MOL=0: the generic
# instantiation): kernel time with the int64 rows + pack, then the counting tests on the product library
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/c
for rep in 1 2; do
echo "generic: $(GSN_COUNT_MOL=0 timeout 300 python scripts/gpu/count_ab.py 2>&1 | grep 'int64 rows True  pack True' | tr '\n' ' ')" | tee -a gpurun_out/c/count_var.log
for so in gsn_amd/lib/libgsn_hip.so gsn_amd/lib/variants/libgsn_hip_c*.so; do
  echo "$(basename $so .so): $(GSN_LIB_PATH=$so timeout 300 python scripts/gpu/count_ab.py 2>&1 | grep 'int64 rows True  pack True' | tr '\n' ' ')" | tee -a gpurun_out/c/count_var.log
done
done
timeout 1500 python -m pytest tests/test_count_gpu.py tests/test_directed_gpu.py tests/test_dataset_gpu.py tests/test_big_batch_gpu.py tests/test_pack16_gpu.py -x -q 2>&1 | tail -3
