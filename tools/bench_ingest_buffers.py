"""From encoded image buffers to HBase rows (the whole extractor leg): N worker processes decode /
bytescale / LANCZOS-resize / crop / mean-subtract into the page-locked ring (extractor/preprocess_pool.py), the GPU runs
DeepSentibank on whole ring batches, the parent normalises + base64-encodes the rows.  Prints images/s with the cores used,
next to the serial path (one Python process preprocessing) and checks that both produce identical rows.
usage: python tools/bench_ingest_buffers.py [n_images] [workers]"""
import io, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from columbiaimagesearch_amd.extractor import GenericExtractor
from columbiaimagesearch_amd.extractor.preprocess_pool import PreprocessPool
from columbiaimagesearch_amd.featurizer.synthetic import sentibank_weights


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    cores = len(os.sched_getaffinity(0))
    quota = "none"
    try:  # the container's CPU time allowance (cgroup v2 cpu.max = "<quota us> <period us>" or "max ...")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = "unlimited" if q == "max" else "%.1f cores" % (float(q) / float(per))
    except Exception:
        pass
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else None
    tmp = tempfile.mkdtemp(prefix="cis_ing_")
    np.savez(os.path.join(tmp, "w.npz"), **sentibank_weights(0))
    np.save(os.path.join(tmp, "mean.npy"), np.zeros((3, 256, 256)) + 110.0)
    ex = GenericExtractor("full", "sbpycaffe", "image", "ext", "EX_", {"EX_sbcaffe_path": os.path.join(tmp, "w.npz"),
                                                                       "EX_imgmean_path": os.path.join(tmp, "mean.npy")})
    rs = np.random.RandomState(0)
    base = []
    for i in range(64):  # 64 distinct JPEGs of web-image size, cycled
        b = io.BytesIO()
        h, w = 300 + 8 * (i % 16), 400 + 16 * (i % 8)
        img = (rs.rand(h // 8, w // 8, 3) * 255).astype(np.uint8)
        Image.fromarray(img).resize((w, h), Image.BICUBIC).save(b, format="JPEG", quality=90)
        base.append(b.getvalue())
    bufs = [base[i % 64] for i in range(n)]
    t = time.perf_counter(); rows_serial = ex.process_batch(bufs[:256]); t_serial = (time.perf_counter() - t) / 256
    pool = PreprocessPool(ex.featurizer, workers=workers, slots=1024)
    workers = pool.workers
    try:
        ex.process_batch(bufs[:1024], pool=pool)  # warm-up: workers import PIL, ring pages get touched
        t = time.perf_counter(); rows = ex.process_batch(bufs, pool=pool); dt = time.perf_counter() - t
        t = time.perf_counter()
        for a in range(0, n, 1024):
            pool.run(bufs[a:a + 1024])
        d_pool = time.perf_counter() - t
        import torch
        ring = pool.ring[:512]
        torch.cuda.synchronize(); t = time.perf_counter()
        x = torch.from_numpy(ring).cuda(non_blocking=True); torch.cuda.synchronize(); d_up = time.perf_counter() - t
        t = time.perf_counter(); f = ex.featurizer.net.forward_dev(x).cpu().numpy(); d_fw = time.perf_counter() - t
        t = time.perf_counter(); [ex._row(f[k]) for k in range(512)]; d_rows = time.perf_counter() - t
        print("stages: workers alone %.0f images/s; per 512 images: upload %.1f ms, forward + read-back %.1f ms, rows %.1f ms"
              % (n / d_pool, d_up * 1e3, d_fw * 1e3, d_rows * 1e3))
    finally:
        pool.close()
    same = all(rows[i] == rows_serial[i] for i in range(256))
    print("buffers -> rows: %d images, %d worker processes on %d visible cores (cgroup CPU allowance: %s), ring page-locked %s: %.0f images/s (serial preprocessing in one process: %.0f images/s); rows identical to the serial path: %s"
          % (n, workers, cores, quota, pool._pinned, n / dt, 1.0 / t_serial, same))


if __name__ == "__main__":  # worker processes are spawned: they re-import this file
    main()
