# the tower products with the library in the tree against a second build (CTRHIP_LIB), same box, same process order.
# The second build: compile the other version of csrc/gemm.hip to an object of its own and link it with the tree's other
# objects into recommendation_amd/lib/exp/libctrhip_base.so (the hipcc lines of csrc/Makefile), here, before the GPU call.
set -e
B=$PWD/recommendation_amd/lib/exp/libctrhip_base.so
for mode in "" dcn; do
  echo "== base $mode"; env F16=1 CTRHIP_LIB=$B timeout -k 10 100 python3 tools/gemm_bench.py $mode | grep -v amdgpu.ids | cut -c1-62 | head -13
  echo "== new $mode";  env F16=1 timeout -k 10 100 python3 tools/gemm_bench.py $mode | grep -v amdgpu.ids | cut -c1-62 | head -13
done
