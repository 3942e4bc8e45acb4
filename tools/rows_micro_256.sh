for g in "8 256 200 336" "8 256 100 168" "8 256 50 84" "8 256 25 42" "64 256 28 28"; do COUT=256 MODES=rows32,ring timeout -k 10 100 python tools/rows_micro.py $g 2>&1 | grep median; done
COUT=512 MODES=rows32,ring timeout -k 10 100 python tools/rows_micro.py 64 256 28 28 2>&1 | grep median
