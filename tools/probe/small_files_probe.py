"""What does a scene's ~2250 small stage files cost on this box's temporary directory?  449 files of 27 KB / 307 KB into a fresh directory:
create + write + close vs write into files created earlier, 1 / 2 / 4 threads, one directory vs one directory per thread."""
import os, sys, tempfile, time, shutil
from concurrent.futures import ThreadPoolExecutor
root = tempfile.mkdtemp(prefix='roreg_small_', dir=os.environ.get('ROREG_BENCH_TMP'))
def write(path, blob, flags):
    fd = os.open(path, flags, 0o644)
    os.write(fd, blob); os.close(fd)
def run(label, n, size, threads, dirs, precreate):
    base = tempfile.mkdtemp(dir=root)
    ds = [f'{base}/d{k}' for k in range(dirs)]
    for d in ds: os.makedirs(d)
    paths = [f'{ds[i % dirs]}/{i // 7}-{i}.npy' for i in range(n)]
    blob = b'x' * size
    t_pre = 0.0
    if precreate:
        t0 = time.perf_counter()
        for p in paths:
            os.close(os.open(p, os.O_CREAT | os.O_WRONLY | os.O_TRUNC, 0o644))
        t_pre = time.perf_counter() - t0
    flags = os.O_WRONLY | (0 if precreate else os.O_CREAT | os.O_TRUNC)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(lambda k: [write(p, blob, flags) for p in paths[k::threads]], range(threads)))
    dt = time.perf_counter() - t0
    print(f'{label:34s} n {n} x {size // 1024:4d} KB, {threads} threads, {dirs} dirs: {1e3 * dt:7.1f} ms = {1e3 * dt / n:.3f} ms per file' + (f'  (+ {1e3 * t_pre:.1f} ms creating them earlier, one thread)' if precreate else ''), flush=True)
    shutil.rmtree(base)
print('directory:', root, ' filesystem:', [l.split()[2] for l in open('/proc/mounts') if l.split()[1] in ('/tmp', '/')][:2])
for size in (27 * 1024, 307 * 1024):
    for threads, dirs in ((1, 1), (2, 1), (4, 1), (4, 4)):
        run('create + write + close', 449, size, threads, dirs, False)
        run('write into existing files', 449, size, threads, dirs, True)
shutil.rmtree(root)
