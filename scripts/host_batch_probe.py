import os, sys, time
sys.path.insert(0, ".")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
from bench import FIXTURES, load_graph_fixture
from redis_hnsw_amd import Index
N, dim, M, ef, k = 1_000_000, 128, 16, 200, 10
V = np.random.default_rng(1).random((N, dim), dtype=np.float32)
Qall = np.random.default_rng(2).random((8192, dim), dtype=np.float32)
g, _ = load_graph_fixture(FIXTURES[(N, dim, M, ef)], V)
ix = Index("t", dim, M, ef); ix.import_graph(g)
def meas(tag):
    Q = Qall[:8192]
    ix.search_batch(Q, k)
    ts = []
    for _ in range(6):
        t = time.perf_counter(); ix.search_batch(Q, k); ts.append(time.perf_counter() - t)
    print(tag, " ".join("%.3f" % (1e3 * x) for x in ts), flush=True)
meas("fresh")
streams = [torch.cuda.Stream() for _ in range(4)]
dev = torch.device("cuda", 0)
dQ = torch.from_numpy(Qall[:1024]).to(dev)
o = (torch.empty((1024, k), dtype=torch.int32, device=dev), torch.empty((1024, k), dtype=torch.float32, device=dev), torch.empty((1024,), dtype=torch.int32, device=dev))
for i in range(60):
    ix.search_batch_device(dQ.data_ptr(), 1024, k, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), streams[i % 3].cuda_stream)
torch.cuda.synchronize()
meas("after device launches on 3 torch streams")
ix.set_tuning("launch_concurrency", 1); ix.set_tuning("waves_per_cu", 4); ix.set_tuning("visited_bounded", 0)
ix.search_batch_device(dQ.data_ptr(), 1024, k, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), 0); torch.cuda.synchronize()
ix.set_tuning("visited_bounded", 1); ix.set_tuning("waves_per_cu", 8); ix.set_tuning("launch_concurrency", 0)
meas("after the exact-counter tuning round trip")
