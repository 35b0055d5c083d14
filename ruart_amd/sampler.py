"""Deterministic batch sampler with the reference's index stream (Utils/VQA_Sampler.py:1-53), plus rank sharding.

Reference behaviour kept: epoch e of the training stream is ``np.random.seed(e + 1333); permutation(indices)``, epochs are
concatenated and cut into consecutive batches; evaluation walks the indices in order and wraps around to fill the last
batch; ``batch_st`` skips already-consumed batches (data-order resume).

Data parallelism (new): with ``world_size`` W every step draws one GLOBAL batch of ``W * batch_size`` indices from that
same stream and rank r takes ``global[r::W]`` - so a W-rank run consumes exactly the samples a single process with
batch size ``W * batch_size`` would, in the same order, and stays resumable with the same ``batch_st``.
"""
import numpy as np
from torch.utils.data import Sampler


class VQA_Sampler(Sampler):
    def __init__(self, source_dt, max_batch_number, batch_size, train, batch_st=None, epoch=None, rank=0, world_size=1):
        self.batch_size = batch_size
        self.data_cnt = len(source_dt)
        self.train = train
        self.rank, self.world = rank, world_size
        gb = batch_size * world_size
        if train:
            self.max_batch_number = int(self.data_cnt * epoch / gb) if epoch is not None else max_batch_number
        else:
            assert epoch is None
            self.max_batch_number = -(-self.data_cnt // gb)
        self.batch_st = batch_st or 0

    def __len__(self):
        return self.max_batch_number

    def __iter__(self):
        gb = self.batch_size * self.world
        pool, epoch = [], 0
        indices = list(range(self.data_cnt))
        for b in range(self.max_batch_number):
            while len(pool) < gb:
                if self.train:
                    np.random.seed(epoch + 1333)
                    pool = pool + np.random.permutation(indices).tolist()
                else:
                    pool = pool + indices
                epoch += 1
            batch, pool = pool[:gb], pool[gb:]
            if b >= self.batch_st:
                yield batch[self.rank::self.world]
