for s in 0 2 3 4 6 8; do echo "== CG_NN_SPLITS=$s"; CG_PAD_SKIP=0 CG_NN_SPLITS=$s python scripts/kbench.py 128 --only b4,b45,dconv2 2>/dev/null | grep "D\."; done
