for v in 0 5 10 20 40 80 200; do echo "skew $v"; for i in 1 2; do
SPACAP_SKEW_US=$v python bench.py --steps 100 --warmup 20 --no-configs --no-in-step --no-drop-in 2>&1 | tail -1 | cut -c150-185; done; done
