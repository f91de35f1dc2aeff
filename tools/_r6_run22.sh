export MELD_DEV=1
for a in 1 0 1 0; do echo -n "async $a: "; MELD_FRAME_ASYNC=$a python bench.py --cpu-sample 0 --no-host-input --no-extra --steps 20 2>/dev/null | python tools/_benchline.py | head -1; done
