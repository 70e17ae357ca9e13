import numpy as np, torch, sys, os, subprocess
sys.path.insert(0, '.')
if len(sys.argv) > 1:
    from roreg_amd import hip
    rng = np.random.default_rng(43)
    s = torch.from_numpy(rng.standard_normal((2500, 32)).astype(np.float32) * 0.5).cuda(); t = torch.from_numpy(rng.standard_normal((2500, 32)).astype(np.float32) * 0.5).cuda()
    seg = hip.Segments([2500])
    import time
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = hip.sinkhorn_batch(s, t, seg, seg, 1.5, 100, recompute=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    np.save(sys.argv[1], out[2].cpu().numpy()); print('ms', dt * 1e3)
else:
    for v in ('0', '6'):
        r = subprocess.run([sys.executable, __file__, f'/tmp/redo_{v}.npy'], env=dict(os.environ, ROREG_OT_FVAR=v), capture_output=True, text=True)
        print(v, r.stdout.strip(), r.stderr[-300:] if r.returncode else '')
    a, b = np.load('/tmp/redo_0.npy'), np.load('/tmp/redo_6.npy')
    print('bitwise equal:', np.array_equal(a, b), 'max diff', np.abs(a - b).max())
