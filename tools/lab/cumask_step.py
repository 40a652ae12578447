"""Lab: partition the chip between the step and the sampling chain with CU-masked streams (hipExtStreamCreateWithCUMask).
   MODE=probe      : where do workgroups of a masked stream land (bit -> (xcc, se, cu) map), also under graph replay
   MODE=step       : the pipelined training step (cfg2) in four arrangements, replay duration of the step's graph (HIP events on
                     its stream) and wall time per step
"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

probe = ctypes.CDLL(os.path.join(ROOT, "tools", "lab", "libcumask_probe.so"))
probe.probe_create_stream.restype = ctypes.c_void_p
probe.probe_create_stream.argtypes = [ctypes.c_void_p, ctypes.c_int]
probe.probe_where.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
probe.probe_spin.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda:0")
torch.cuda.init()
torch.zeros(1, device=dev)


def masked_stream(bits):
    """bits: iterable of enabled bit indices (0..255)"""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    h = probe.probe_create_stream(ctypes.cast(words, ctypes.c_void_p), 8)
    assert h, "hipExtStreamCreateWithCUMask failed"
    return torch.cuda.ExternalStream(h, device=dev)


def where(stream, grid, threads=256, lds=0):
    out = torch.full((grid, 2), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    rc = probe.probe_where(ctypes.c_void_p(stream.cuda_stream), grid, threads, lds, ctypes.c_void_p(out.data_ptr()))
    assert rc == 0, rc
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype("int64") & 0xFFFFFFFF
    xcc = o[:, 0] & 0xF
    hw = o[:, 1]
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 0x7
    return [(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(xcc, se, sh, cu)]


def mode_probe():
    full = torch.cuda.Stream(device=dev)
    w = where(full, 512)
    print("unmasked: distinct (xcc,se,sh,cu) =", len(set(w)), "; per xcc:", sorted({x: sum(1 for t in set(w) if t[0] == x) for x in range(8)}.items()))
    print("  first 16 blocks:", w[:16])
    for b in (0, 1, 7, 8, 9, 16, 32, 33, 64, 255):
        s = masked_stream([b])
        print(f"bit {b:3d} ->", sorted(set(where(s, 16))))
    s8 = masked_stream(range(8))
    print("bits 0..7, grid 8 x 1024 threads:", where(s8, 8, 1024))
    rest = masked_stream(range(8, 256))
    w = where(rest, 512)
    print("bits 8..255: distinct =", len(set(w)), "; overlap with bits 0..7:", sorted(set(w) & set(where(s8, 64))))
    # under graph replay
    out = torch.full((512, 2), -1, dtype=torch.int32, device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s8):
        probe.probe_where(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 64, 256, 0, ctypes.c_void_p(out.data_ptr()))
    torch.cuda.synchronize()
    out.fill_(-1)
    with torch.cuda.stream(s8):
        g.replay()
    torch.cuda.synchronize()
    o = out[:64].cpu().numpy().astype("int64") & 0xFFFFFFFF
    print("graph captured on the 8-CU stream, replayed on it: distinct CUs =", len({(int(a) & 0xF, (int(b) >> 8) & 0xFFF) for a, b in o}))
    out.fill_(-1)
    with torch.cuda.stream(full):
        g.replay()
    torch.cuda.synchronize()
    o = out[:64].cpu().numpy().astype("int64") & 0xFFFFFFFF
    print("same graph replayed on an unmasked stream: distinct CUs =", len({(int(a) & 0xF, (int(b) >> 8) & 0xFFF) for a, b in o}))
    # a fork inside a captured graph: which CUs does the forked branch get?
    out2 = torch.full((64, 2), -1, dtype=torch.int32, device=dev)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s8):
        cur = torch.cuda.current_stream()
        probe.probe_where(ctypes.c_void_p(cur.cuda_stream), 64, 256, 0, ctypes.c_void_p(out.data_ptr()))
        full.wait_stream(cur)
        probe.probe_where(ctypes.c_void_p(full.cuda_stream), 64, 256, 0, ctypes.c_void_p(out2.data_ptr()))
        probe.probe_where(ctypes.c_void_p(cur.cuda_stream), 64, 256, 0, ctypes.c_void_p(out[64:].data_ptr()))
        cur.wait_stream(full)
    torch.cuda.synchronize()
    out.fill_(-1); out2.fill_(-1)
    with torch.cuda.stream(s8):
        g2.replay()
    torch.cuda.synchronize()
    def ncu(t):
        o = t.cpu().numpy().astype("int64") & 0xFFFFFFFF
        return len({(int(a) & 0xF, (int(b) >> 8) & 0xFFF) for a, b in o})
    print("forked capture (8-CU origin stream + unmasked fork), replayed on the 8-CU stream: origin branch CUs =", ncu(out[:128]),
          " forked branch CUs =", ncu(out2))


def mode_step():
    from spacap3d_amd import engine, synthetic as S
    from spacap3d_amd import detector
    from spacap3d_amd.engine import Trainer, synthetic_batch
    from spacap3d_amd.spacapnet import build_default
    arrangement = os.environ.get("ARR", "base")
    nres = int(os.environ.get("NRES", "8"))
    res_bits = list(range(nres))            # one CU per XCD (bit b -> xcc b % 8)
    if arrangement != "base":
        s_main = masked_stream(range(nres, 256))
        s_fps = masked_stream(res_bits)
        s_rest = torch.cuda.Stream(device=dev)
        key = ("cuda", 0)
        engine._STREAMS[key] = {"side": s_fps if arrangement in ("split", "splitgraph", "all8") else torch.cuda.Stream(device=dev),
                                "capture": s_main, "comm": torch.cuda.Stream(device=dev)}
    if arrangement in ("split", "splitgraph"):
        from spacap3d_amd import pointnet2_utils as pu
        from spacap3d_amd.pointnet2_modules import PointnetFPModule
        from spacap3d_amd.sa_mlp import rows_index

        def pyramid_split(xyz, npoints=detector.SA_NPOINTS, radii=detector.SA_RADII, nsamples=detector.SA_NSAMPLES):
            cur_s = torch.cuda.current_stream(dev)
            inds_all, idx_all, xyzs, evs = [], [], [xyz], []
            cur = xyz
            for n in npoints:
                inds = pu.furthest_point_sample(cur, n)
                new_xyz = detector._centres(cur, inds)
                inds_all.append(inds); xyzs.append(new_xyz); cur = new_xyz
                ev = torch.cuda.Event(); ev.record(cur_s); evs.append(ev)
            with torch.cuda.stream(s_rest):
                for l, (r, ns) in enumerate(zip(radii, nsamples)):
                    s_rest.wait_event(evs[l])
                    idx_all.append(pu.ball_query(r, ns, xyzs[l], xyzs[l + 1]))
                fp1 = PointnetFPModule.neighbours(xyzs[3], xyzs[4])
                fp2 = PointnetFPModule.neighbours(xyzs[2], xyzs[3])
                out = tuple(inds_all) + tuple(idx_all) + (fp1[0], fp1[1], fp2[0], fp2[1])
                out = out + tuple(rows_index(idx_all[l], xyzs[l].shape[1]) for l in (1, 2, 3))
                out = out + tuple(x.contiguous() for x in xyzs[1:])
            cur_s.wait_stream(s_rest)
            if not torch.cuda.is_current_stream_capturing():
                for t in out:
                    t.record_stream(s_rest); t.record_stream(cur_s)
            return out
        engine.geometry_pyramid = pyramid_split
        if arrangement == "split":
            os.environ["SPACAP_PREFETCH_GRAPH"] = "0"
    torch.manual_seed(0)
    model = build_default(input_feature_dim=1, num_proposal=256).to(dev).train()
    trainer = Trainer(model, S.mean_size_arr().numpy(), use_relation=True)
    if arrangement != "base" or os.environ.get("NORESERVE"):
        # the mask does the reserving (grids follow SPACAP_LAB_CUS): a preset side stream keeps prefetch() from reserving CUs
        trainer.side_stream = engine._role_stream(dev, "side")
    datas = [synthetic_batch(8, 40000, dev, seed=1000 + i) for i in range(2)]
    trainer.step(datas[0], next_data=datas[1])
    assert trainer.enable_graph(datas[1]), trainer.graph_error
    # time the graph replay alone on its stream
    g = trainer.graph
    class Timed:
        def __init__(self): self.ev = []
        def replay(self):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record(); self.ev.append((a, b))
    tg = Timed(); trainer.graph = tg
    def run(n, nd=True):
        tg.ev.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            trainer.step(datas[i % 2], next_data=datas[(i + 1) % 2] if nd else None)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / n * 1e3
        rep = sum(a.elapsed_time(b) for a, b in tg.ev) / len(tg.ev)
        return wall, rep
    run(20)
    wall, rep = run(40)
    print(f"ARR={arrangement} NRES={nres} LAB_CUS={os.environ.get('SPACAP_LAB_CUS')}: pipelined  wall {wall:.3f} ms/step, graph replay {rep:.3f} ms", flush=True)
    # no side-stream work: pyramid computed once and re-attached
    trainer.prefetch(datas[0]); torch.cuda.synchronize()
    saved = datas[0]["_fps_prefetch"]
    tg.ev.clear()
    def reuse(n):
        tg.ev.clear()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            datas[0]["_fps_prefetch"] = saved
            trainer.step(datas[0], next_data=None)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / n * 1e3
        return wall, sum(a.elapsed_time(b) for a, b in tg.ev) / len(tg.ev)
    reuse(10)
    wall, rep = reuse(40)
    print(f"ARR={arrangement}: no side-stream work   wall {wall:.3f} ms/step, graph replay {rep:.3f} ms", flush=True)


if __name__ == "__main__":
    {"probe": mode_probe, "step": mode_step}[os.environ.get("MODE", "probe")]()
