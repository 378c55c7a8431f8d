export TMPDIR=/tmp
for sk in -1 1 2 4 6 9 18; do echo "splitk $sk"; python3 tools/sweep_conv.py --batch 16 --cfgs=-1,2,3 --only ssd_b --splitk $sk --iters 50 2>/dev/null | grep -v "^layer"; done
