"""Where does the evaluator's engine route spend its input / output time?  60 x 38.4 MB files in /tmp: reads into pinned memory on 1 / 4 / 8 threads
(first and second time: the pinned allocator's cache), uploads, downloads into pinned memory, np.save."""
import os, sys, time, tempfile, shutil
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
root = tempfile.mkdtemp(prefix='io_probe_')
n = 60
a = np.random.rand(5000, 32, 60).astype(np.float32)
for i in range(n):
    np.save(f'{root}/{i}.npy', a)
torch.cuda.init(); torch.zeros(1).cuda()
def read(i):
    with open(f'{root}/{i}.npy', 'rb') as f:
        v = np.lib.format.read_magic(f); shape, fo, dt = np.lib.format.read_array_header_1_0(f)
        t0 = time.perf_counter()
        dst = torch.empty(shape, dtype=torch.float32, pin_memory=True)
        t1 = time.perf_counter()
        buf = memoryview(dst.numpy()).cast('B'); got = 0
        while got < len(buf):
            got += f.readinto(buf[got:])
        return dst, t1 - t0, time.perf_counter() - t1
for threads in (1, 4, 8, 8):
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as pool:
        res = list(pool.map(read, range(n)))
    dt = time.perf_counter() - t0
    print(f'read {n} files on {threads} threads: {dt:.3f} s = {n * a.nbytes / dt / 1e9:.1f} GB/s; pinned alloc total {sum(r[1] for r in res):.3f} s, readinto total {sum(r[2] for r in res):.3f} s (thread-seconds)')
    hosts = [r[0] for r in res]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev = [h.to('cuda', non_blocking=True) for h in hosts]
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'   upload: enqueue {t1 - t0:.3f} s, done {t2 - t0:.3f} s = {n * a.nbytes / (t2 - t0) / 1e9:.1f} GB/s')
    t0 = time.perf_counter()
    outs = []
    for d in dev:
        h = torch.empty(d.shape, dtype=d.dtype, pin_memory=True); h.copy_(d, non_blocking=True); outs.append(h)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'   download into fresh pinned tensors: enqueue {t1 - t0:.3f} s, done {t2 - t0:.3f} s = {n * a.nbytes / (t2 - t0) / 1e9:.1f} GB/s')
    t0 = time.perf_counter()
    with ThreadPoolExecutor(4) as pool:
        list(pool.map(lambda q: np.save(f'{root}/out_{q}.npy', outs[q].numpy()), range(n)))
    print(f'   np.save x {n} on 4 threads: {time.perf_counter() - t0:.3f} s')
    del res, hosts, dev, outs
# the pageable path the first version used
t0 = time.perf_counter()
dev = [torch.from_numpy(np.load(f'{root}/{i}.npy', mmap_mode='r')).to('cuda') for i in range(n)]
torch.cuda.synchronize(); print(f'mmap + pageable .to(cuda): {time.perf_counter() - t0:.3f} s')
t0 = time.perf_counter()
dev = [torch.from_numpy(np.load(f'{root}/{i}.npy')).to('cuda') for i in range(n)]
torch.cuda.synchronize(); print(f'np.load + pageable .to(cuda): {time.perf_counter() - t0:.3f} s')
shutil.rmtree(root)
