import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_config3 as T
for prec in sys.argv[1:]:
    L, ok = T._apollo_losses(prec, 148, 4)
    print(prec, ok, {k: round(v, 5) for k, v in L.items()}, flush=True)
