"""CPU stand-in engine for bench.py's N > 1 control flow (selected with WMAR_BENCH_ENGINE=tests.bench_stub:make and
WMAR_BENCH_BACKEND=gloo by tests/test_bench_distributed_cpu.py).  It has the wrapper / watermarker methods bench.py calls and
nothing else: no model, no kernels.  Test infrastructure only."""
import torch


class StubModel:
    device = torch.device("cpu")

    def set_watermarker(self, wm):
        self.wm = wm

    def sample(self, cond, gen_params, apply_watermark=False):
        c = torch.as_tensor(cond).view(-1, 1)
        return (c * 7 + torch.randint(0, 1000, (c.shape[0], 256))) % 16384

    def codes_to_images(self, codes):
        return codes.float().view(-1, 1, 16, 16).repeat(1, 3, 1, 1) / 8192.0 - 1.0

    def images_to_codes(self, images):
        return ((images[:, 0] + 1.0) * 8192.0).round().long().view(-1, 256)


class StubWatermark:
    def __init__(self, rank):
        self.rank = rank
        self.table = None

    def key_table(self):
        if self.table is None:
            self.table = torch.arange(64, dtype=torch.int32).view(8, 8) + 100 * (self.rank + 1)   # rank-dependent on purpose
        return self.table

    def set_key_table(self, table):
        self.table = table

    def detect_counts(self, codes):
        n = codes.shape[0]
        tag = int(self.table[0, 0])          # every rank must hold rank 0's table after the broadcast: 100
        return (torch.full((n,), 0.5, dtype=torch.float64), torch.full((n,), 255, dtype=torch.int32),
                torch.full((n,), tag, dtype=torch.int32))


def make(device, rank, args):
    return StubModel(), StubWatermark(rank), {}
