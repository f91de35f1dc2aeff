export MELD_DEV=1
for m in 7 6; do echo "== EE max KB $m"; MELD_KNN16_EE_MAXKB=$m python bench.py --cells 1000000 --dims 100 --steps 3 --no-extra --cpu-sample 0 --no-host-input --stages 2>/dev/null | python tools/_benchline.py; done
