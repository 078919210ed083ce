"""Minimal G+D training step around the WC generator: hinge loss, Adam(beta1=0, beta2=0.9)
(run.py:58-59,255-256), `training_ratio` critic updates per generator update (run.py:101), the
generator update at batch_size x generator_batch_multiple (run.py:293-294).

Data parallelism is new design (the reference is single-process): one process per
GPU, per-replica WC statistics by default, and ONE flat gradient bucket per network that is
all-reduced (mean) over RCCL after each backward.  The bucket is zero-copy: every parameter's .grad
is a view into it, so there is no pack/unpack around the collective.  xGMI is point-to-point
(7 links per GPU) and these messages are small (G ~19 MB, D ~4 MB fp32), i.e. latency-bound; a single
large all-reduce per network is the shape RCCL handles best there.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import _state


# development: run the gradient collectives even in a one-rank process group (exercises RCCL between graph segments on one GPU)
_FORCE_COLLECTIVES = __import__('os').environ.get('WC_FORCE_COLLECTIVES', '0') == '1'


class FlatGradBucket:
    """All gradients of one network in one contiguous buffer; .grad tensors are views into it.
    `flat=False` (what the trainer picks for a single process): no buffer -- zero() just drops the gradients, so autograd
    hands each parameter its gradient tensor instead of adding it into a view (one small kernel per parameter and
    backward pass saved; nothing is all-reduced anyway)."""

    def __init__(self, params, process_group=None, flat=True):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.flat = None
        if not flat:
            return
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            seg = self.flat[off:off + p.numel()]
            if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
                o, i, kh, kw = p.shape                      # same strides as the channels_last weight
                p.grad = seg.view(o, kh, kw, i).permute(0, 3, 1, 2)
            else:
                p.grad = seg.view_as(p)
            off += p.numel()

    def zero(self):
        if self.flat is None:
            for p in self.params:
                p.grad = None
        else:
            self.flat.zero_()

    def allreduce_mean(self):
        if (self.world > 1 or _FORCE_COLLECTIVES) and self.flat is not None:
            tm = self.timer
            if tm is not None:
                tm.begin(self.flat)
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            if tm is not None:
                tm.end(self.flat)
            self.flat.mul_(1.0 / self.world)

    timer = None          # an AllreduceTimer while bench.py measures the collectives' share of a step (diagnostics only)


class AllreduceTimer:
    """Time spent in the gradient all-reduces, for the N > 1 diagnostics of bench.py: HIP events on the launching stream
    around each collective (the NCCL/RCCL call makes that stream wait for its result), wall clock on the CPU (gloo)."""

    def __init__(self):
        self.pairs, self.cpu_s, self.calls = [], 0.0, 0

    def begin(self, t):
        self.calls += 1
        if t.is_cuda:
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
            self._e0 = e0
        else:
            import time
            self._t0 = time.perf_counter()

    def end(self, t):
        if t.is_cuda:
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
            self.pairs.append((self._e0, e1))
        else:
            import time
            self.cpu_s += time.perf_counter() - self._t0

    def total_ms(self):
        if self.pairs:
            torch.cuda.synchronize()
            return sum(a.elapsed_time(b) for a, b in self.pairs)
        return self.cpu_s * 1e3


def broadcast_state(module, src=0, group=None):
    """Rank 0's parameters and buffers (incl. moving_mean / moving_cov) to every replica at start-up."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def _bump_versions(params):
    """torch's fused Adam updates the weights without moving their version counters; everything that caches per
    `tensor._version` (the eval-mode WC plan keys on the coloring weights) must still see the update."""
    inc = getattr(torch.autograd.graph, 'increment_version', None)
    if inc is not None:
        for p in params:
            inc(p)


class GanTrainer:
    def __init__(self, generator, discriminator, batch_size=64, generator_batch_multiple=2, training_ratio=5,
                 lr=2e-4, beta1=0.0, beta2=0.9, noise_dim=128, number_of_classes=10, conditional=False,
                 process_group=None, seed=1234, flat_buckets=None):
        self.G, self.D = generator, discriminator
        self.batch_size, self.gbm, self.training_ratio = batch_size, generator_batch_multiple, training_ratio
        self.noise_dim, self.K, self.conditional = noise_dim, number_of_classes, conditional
        self.dev = next(generator.parameters()).device
        world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        flat = (world > 1) if flat_buckets is None else bool(flat_buckets)     # (True at world 1: the DP layout, for tests)
        self.g_bucket = FlatGradBucket(self.G.parameters(), process_group, flat=flat)
        self.d_bucket = FlatGradBucket(self.D.parameters(), process_group, flat=flat)
        # capturable: the step counters live on the device, so a whole G+D step can be recorded into one hipGraph
        cap = self.dev.type == 'cuda'
        # fused: one multi-tensor kernel per update instead of ~10 small launches per parameter (step counters on the
        # device either way)
        kw = dict(fused=True) if cap else dict()
        self.opt_g = torch.optim.Adam(self.g_bucket.params, lr=lr, betas=(beta1, beta2), capturable=cap, **kw)
        self.opt_d = torch.optim.Adam(self.d_bucket.params, lr=lr, betas=(beta1, beta2), capturable=cap, **kw)
        if cap:
            # the default device generator (its Philox offset is graph-safe); every replica draws its own noise
            rank = dist.get_rank(process_group) if (dist.is_available() and dist.is_initialized()) else 0
            torch.cuda.manual_seed(seed + rank)
        self._graph = None
        self._side = None
        self.overlap_g_forward = True
        self._boundary = None            # set while capture_segments() records: called where a gradient all-reduce goes
        self._side_pending = False

    def _sync_grads(self, bucket):
        """The gradient all-reduce of one network (a no-op for a single process) -- or, while capture_segments() records,
        the end of one graph segment and the start of the next."""
        if self._boundary is not None:
            self._boundary(bucket)
        else:
            bucket.allreduce_mean()

    def _noise(self, n):
        z = torch.randn(n, self.noise_dim, device=self.dev)
        cls = torch.randint(0, self.K, (n, 1), device=self.dev, dtype=torch.int32)
        return z, cls

    def _d(self, x, cls):
        out = self.D(x, cls if self.conditional else None)
        return out[0] if isinstance(out, tuple) else out

    def generate(self, rounds):
        """The `rounds` generated batches of one G+D step in ONE generator pass.  The generator's weights are fixed
        while the critic trains, so its `rounds` forward passes are independent; they run as a single batch of
        rounds x batch_size whose WC layers keep per-round statistics (layers.statistic_groups) -- the same numbers
        as separate passes, with the covariance / Cholesky problems of the rounds solved side by side."""
        from .layers import statistic_groups, supports_statistic_groups
        z, cls = self._noise(self.batch_size * rounds)
        if rounds > 1 and not supports_statistic_groups(self.G):
            # a norm layer without the grouped form (zca, renorm, padded widths, plain batch norm): `rounds` real passes
            with torch.no_grad():
                fakes = [self.G(zz, cc) for zz, cc in zip(z.split(self.batch_size), cls.split(self.batch_size))]
            return fakes, cls.split(self.batch_size)
        with torch.no_grad(), statistic_groups(rounds):
            fake = self.G(z, cls)                      # train-mode WC forward (batch statistics), no graph
        return fake.split(self.batch_size), cls.split(self.batch_size)

    def d_step(self, real, real_cls=None, fake=None, cls=None):
        if fake is None:
            (fake,), (cls,) = self.generate(1)
        self.d_bucket.zero()
        # real and generated images go through the critic as ONE batch of 128: the critic has no batch-dependent
        # layer (discriminator_norm is 'n' in every recipe), so this equals two applications with shared weights,
        # with one spectral-norm power iteration per update and convolutions at twice the batch
        n = real.shape[0]
        if self.conditional:
            if real_cls is None:      # run.py:317 get_dataset(supervised=True): real images come with their labels
                raise ValueError("conditional critic: d_step needs the labels of the real batch (real_cls)")
            both_cls = torch.cat([real_cls.reshape(-1, 1).to(cls.dtype), cls], dim=0)
        else:
            both_cls = None
        out = self._d(torch.cat([real, fake], dim=0), both_cls)
        loss = F.relu(1.0 - out[:n]).mean() + F.relu(1.0 + out[n:]).mean()
        loss.backward()
        self._sync_grads(self.d_bucket)
        self.opt_d.step()
        _bump_versions(self.d_bucket.params)
        return loss.detach()

    def g_step(self, generated=None):
        self.g_bucket.zero()
        if generated is None:
            z, cls = self._noise(self.batch_size * self.gbm)
            fake = self.G(z, cls)
        else:
            fake, cls = generated
        for p in self.d_bucket.params:
            p.requires_grad_(False)
        loss = -self._d(fake, cls).mean()
        loss.backward()
        for p in self.d_bucket.params:
            p.requires_grad_(True)
        self._sync_grads(self.g_bucket)
        self.opt_g.step()
        _bump_versions(self.g_bucket.params)
        return loss.detach()

    def step(self, real_batches, real_labels=None):
        """One G+D step: training_ratio critic updates, then one generator update.  `real_labels`: one int (64,) or
        (64, 1) tensor per real batch -- required by the conditional recipes (the projection critic must see true
        (image, label) pairs; run.py:317)."""
        if self.conditional and real_labels is None:
            raise ValueError("conditional recipe: step() needs real_labels (one label tensor per real batch)")
        fakes, clss = self.generate(self.training_ratio)
        generated = None
        if self.overlap_g_forward and self.dev.type == 'cuda':
            # The generator update's forward pass needs the generator's weights only, and those do not move while the
            # critic trains: it runs on a second stream next to the critic updates and fills the chip while their
            # narrow kernels (Cholesky, small GEMMs, spectral norm) run -- and the other way round.  Same numbers.
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream()
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                z, cls = self._noise(self.batch_size * self.gbm)
                generated = (self.G(z, cls), cls)
            self._side_pending = True
        for r in range(self.training_ratio):
            rl = real_labels[r % len(real_labels)] if real_labels is not None else None
            d_loss = self.d_step(real_batches[r % len(real_batches)], real_cls=rl, fake=fakes[r], cls=clss[r])
        if generated is not None:
            if self._side_pending:                   # (a graph segment that ended in between has joined it already)
                torch.cuda.current_stream().wait_stream(self._side)
                self._side_pending = False
            generated[0].record_stream(torch.cuda.current_stream())
            generated[1].record_stream(torch.cuda.current_stream())
        g_loss = self.g_step(generated)
        return d_loss, g_loss

    def trace_boundaries(self, real_batches, real_labels=None):
        """One eager step that also records where the gradient all-reduces sit: ['d', ..., 'd', 'g'] -- the cut points of
        capture_segments().  Every rank must report the same list (tests/test_dp_gloo.py, bench.py --dry-run)."""
        order = []

        def boundary(bucket):
            order.append('d' if bucket is self.d_bucket else 'g')
            bucket.allreduce_mean()

        self._boundary = boundary
        try:
            losses = self.step(real_batches, real_labels)
        finally:
            self._boundary = None
        return order, losses

    def capture(self, real_batches, real_labels=None, warmup=3):
        """Record one whole G+D step (~1700 kernel launches) into ONE hipGraph -- single process only: capturing an RCCL
        all-reduce aborts the process (tried with a one-rank group), so with gradient collectives use capture_segments().

        The C-ABI stages never synchronise or allocate and the gate of the fast path is a device flag, so the
        step is capturable as is; `real_batches` (and `real_labels`, for the conditional recipes) become the graph's
        static inputs: copy new data into them.  Returns a callable that replays the step.  Every replay draws fresh
        noise and updates the weights."""
        if (self.g_bucket.world > 1 or _FORCE_COLLECTIVES) and self.g_bucket.flat is not None:
            raise RuntimeError("capture(): gradient all-reduces cannot be captured; use capture_segments()")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.step(real_batches, real_labels)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._static_losses = self.step(real_batches, real_labels)
        self._graph = graph

        def replay():
            graph.replay()
            _state.replays += 1                      # caches keyed by tensor versions: see _state.py
            return self._static_losses
        return replay

    def capture_segments(self, real_batches, real_labels=None, warmup=3):
        """The step as a CHAIN of hipGraphs cut at the gradient all-reduces, which stay ordinary RCCL calls between the
        replays: no collective is ever captured, yet the ~2000 kernel launches of a step leave the host loop -- the
        eager loop is at the edge of host-bound.  Segments:
        [generated batches + first critic pass] | [critic update + next pass] x (training_ratio - 1) |
        [critic update + generator pass] | [generator update]; the generator forward of the update still forks onto the
        second stream, but joins at the first cut (a captured graph must end with its branches joined).  The graphs share
        one memory pool and are replayed in capture order; autograd state crosses the cuts as ordinary tensors of that
        pool.  Returns a callable that replays the chain.

        If recording fails, the open capture is closed (the stream leaves capture mode), the partial graphs and their
        pool are dropped and the exception is re-raised: the caller can then run eagerly.  With several ranks the
        caller has to AGREE on the mode afterwards (bench.py all-reduces an ok flag): a rank that replays graphs while
        another runs eagerly would still issue the same collectives in the same order, but only by construction of
        step() -- do not rely on it."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.step(real_batches, real_labels)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        segments, cur = [], {}

        def begin():
            g = torch.cuda.CUDAGraph()
            # thread_local: calls of OTHER threads (the process group's watchdog polls its events) must not invalidate
            # a capture that contains none of their work
            ctx = torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local")
            ctx.__enter__()
            cur['g'], cur['ctx'] = g, ctx

        def end(bucket):
            if self._side_pending:                   # join the forked generator forward before the graph ends
                torch.cuda.current_stream().wait_stream(self._side)
                self._side_pending = False
            ctx = cur.pop('ctx')
            ctx.__exit__(None, None, None)
            segments.append((cur.pop('g'), bucket))

        def boundary(bucket):
            # no collective while recording: the captured kernels have not run, there is nothing to reduce -- and a rank
            # whose recording fails half-way must not leave its peers inside an all-reduce
            end(bucket)
            begin()

        self._boundary = boundary
        try:
            begin()
            losses = self.step(real_batches, real_labels)
            end(None)
        except BaseException as exc:
            ctx = cur.pop('ctx', None)
            if ctx is not None:                      # leave capture mode, as the context manager does on an error
                try:
                    ctx.__exit__(type(exc), exc, exc.__traceback__)
                except Exception:
                    pass
            self._side_pending = False
            segments.clear()
            cur.clear()
            del pool
            raise
        finally:
            self._boundary = None
        self._segments = segments

        def replay():
            for g, bucket in segments:
                g.replay()
                if bucket is not None:
                    bucket.allreduce_mean()
            _state.replays += 1                      # caches keyed by tensor versions: see _state.py
            return losses
        return replay


CIFAR10_UNCOND = dict(          # scripts/cifar10_resnet_sn_uncond.sh:4-7 + run.py:147-193
    generator=dict(block_sizes=(256, 256, 256), resamples=("UP", "UP", "UP"), first_block_shape=(4, 4, 256),
                   number_of_classes=10, block_norm='d', block_after_norm='uconv', last_norm='d',
                   last_after_norm='uconv', gan_type=None),
    discriminator=dict(input_image_shape=(32, 32, 3), block_sizes=(128, 128, 128, 128),
                       resamples=('DOWN', 'DOWN', 'SAME', 'SAME'), number_of_classes=10, type=None, spectral=True,
                       sum_pool=True, conv_singular=False),
    image_shape=(32, 32, 3), conditional=False)

CIFAR10_COND = dict(            # scripts/cifar10_resnet_sn_cond.sh:5-8
    generator=dict(block_sizes=(128, 128, 128), resamples=("UP", "UP", "UP"), first_block_shape=(4, 4, 128),
                   number_of_classes=10, block_norm='d', block_after_norm='ucconv', last_norm='d',
                   last_after_norm='uconv', gan_type='PROJECTIVE'),
    discriminator=dict(input_image_shape=(32, 32, 3), block_sizes=(256, 256, 256, 256),
                       resamples=('DOWN', 'DOWN', 'SAME', 'SAME'), number_of_classes=10, type='PROJECTIVE', spectral=True,
                       sum_pool=True, conv_singular=False),
    image_shape=(32, 32, 3), conditional=True)


# scripts/stl10_resnet_sn_uncond.sh:4-7 (generator_filters 256, discriminator_filters 128); run.py:152 first_block_w = 6,
# run.py:165-166 three UP blocks, run.py:333 images 48 x 48 x 3.  WC sites (N = 128): 6, 12, 12, 24, 24, 48, 48(final) at C = 256
STL10_UNCOND = dict(
    generator=dict(block_sizes=(256, 256, 256), resamples=("UP", "UP", "UP"), first_block_shape=(6, 6, 256),
                   number_of_classes=10, block_norm='d', block_after_norm='uconv', last_norm='d',
                   last_after_norm='uconv', gan_type=None),
    discriminator=dict(input_image_shape=(48, 48, 3), block_sizes=(128, 128, 128, 128),
                       resamples=('DOWN', 'DOWN', 'SAME', 'SAME'), number_of_classes=10, type=None, spectral=True,
                       sum_pool=True, conv_singular=False),
    image_shape=(48, 48, 3), conditional=False)

# scripts/tinyimagenet_resnet_sn_cond_sa.sh:4-7 (generator_filters 128, discriminator_filters 1024, ufconv, filters_emb 15,
# PROJECTIVE); run.py:155-158 four UP blocks, run.py:172-173 200 classes, run.py:205-208 five critic blocks
# (filters/4, /2, 1, 1, 1; DOWN x3, SAME x2), run.py:335 images 64 x 64 x 3.  WC sites: 4, 8, 8, 16, 16, 32, 32, 64, 64(final)
# at C = 128; 200 classes > 64 samples per batch, so the block sites run per-sample coloring tables.
TINYIMAGENET_COND_SA = dict(
    generator=dict(block_sizes=(128, 128, 128, 128), resamples=("UP", "UP", "UP", "UP"), first_block_shape=(4, 4, 128),
                   number_of_classes=200, block_norm='d', block_after_norm='ufconv', filters_emb=15, last_norm='d',
                   last_after_norm='uconv', gan_type='PROJECTIVE'),
    discriminator=dict(input_image_shape=(64, 64, 3), block_sizes=(256, 512, 1024, 1024, 1024),
                       resamples=('DOWN', 'DOWN', 'DOWN', 'SAME', 'SAME'), number_of_classes=200, type='PROJECTIVE',
                       spectral=True, sum_pool=True, conv_singular=False, filters_emb=15),
    image_shape=(64, 64, 3), conditional=True)

# the four GPU configurations of BASELINE.json:configs, by the name bench.py --config takes
CONFIGS = {'cifar10_uncond': CIFAR10_UNCOND, 'cifar10_cond': CIFAR10_COND, 'stl10_uncond': STL10_UNCOND,
           'tinyimagenet_cond_sa': TINYIMAGENET_COND_SA}


def wc_sites(config, batch):
    """(name, N, H, W, C) of every WC site of the config's generator at batch size `batch` (bn1 on the
    block input, bn2 after the upsampling conv1, then the final site generator.py:154)."""
    g = config['generator']
    h, w, c = g['first_block_shape']
    sites = []
    for i, (bs, rs) in enumerate(zip(g['block_sizes'], g['resamples'])):
        sites.append((f'Generator.{i}.bn1', batch, h, w, c))
        if rs == 'UP':
            h, w = 2 * h, 2 * w
        c = int(bs)
        sites.append((f'Generator.{i}.bn2', batch, h, w, c))
    sites.append(('Generator.BN.Final', batch, h, w, c))
    return sites


def build_trainer(config=CIFAR10_UNCOND, device='cuda', process_group=None, sync_wc=False, **kw):
    from .discriminator import make_discriminator
    from .generator import make_generator
    G = make_generator(process_group=process_group if sync_wc else None, **config['generator']).to(device)
    D = make_discriminator(**config['discriminator']).to(device)
    broadcast_state(G, group=process_group)
    broadcast_state(D, group=process_group)
    return GanTrainer(G, D, number_of_classes=config['generator']['number_of_classes'],
                      conditional=config['conditional'], process_group=process_group, **kw)
