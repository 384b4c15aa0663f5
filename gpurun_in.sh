python -m pytest tests/test_spconv_gpu.py tests/test_backbone_gpu.py tests/test_bn_gpu.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do python bench.py --steps 300 --cpu-clouds 0 --no-roofline 2>&1 | tail -1 | cut -c80-190; done
echo "mode3"; python bench.py --steps 300 --cpu-clouds 0 --no-roofline --prefetch 3 2>&1 | tail -1 | cut -c80-190
