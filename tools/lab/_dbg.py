import sys, torch, collections
sys.path.insert(0, '.')
from spacap3d_amd import _native
from spacap3d_amd import synthetic as S
from spacap3d_amd.engine import Trainer, synthetic_batch
from spacap3d_amd.spacapnet import build_default
lib = _native.lib
cnt = collections.Counter()
for name in ("spacap_sa_rows_scatter_f32", "spacap_sa_rows_gather_f32", "spacap_sa_rows_index_f32"):
    orig = getattr(lib, name)
    def mk(orig, name):
        def f(*a):
            cnt[name] += 1
            return orig(*a)
        return f
    setattr(lib, name, mk(orig, name))
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_default().to(dev).train()
tr = Trainer(model, S.mean_size_arr().numpy())
data = synthetic_batch(8, 40000, dev, seed=1000)
for i in range(3):
    cnt.clear()
    tr.step(data, next_data=data)
    torch.cuda.synchronize()
    print(i, dict(cnt), len(data.get("fps_pyramid") or ()))
