export TMPDIR=/tmp
for l in conv4_2 b4_trio conv3_1 fc7_full; do
echo "== $l"; RON_STAMPS=1 python3 tools/sweep_conv.py --cfgs=0 --only $l 2>&1 | grep -v "amdgpu.ids\|^layer"
done
echo "== 8 wave"
for l in conv4_2 conv3_1; do
RON_IGEMM256_V1=1 python3 tools/sweep_conv.py --cfgs=0 --only $l 2>&1 | grep -v "amdgpu.ids\|^layer"
done
timeout 600 python3 -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -3
