for i in 1 2 3; do 
REL_NORESERVE=1 python bench.py --steps 200 --warmup 20 --no-configs --no-in-step 2>&1 | tail -1 | cut -c150-185
python bench.py --steps 200 --warmup 20 --no-configs --no-in-step 2>&1 | tail -1 | cut -c150-185; done
