for be in 8 16 32 64; do for bs in 0 8 32; do
echo -n "every=$be slack=$bs: "; MELD_KNN16_BATCH_EVERY=$be MELD_KNN16_BATCH_SLACK=$bs python tools/knn_only.py 1000000 2 2>&1 | tail -1 | cut -c1-40
done; done
