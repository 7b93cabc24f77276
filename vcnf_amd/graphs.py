"""HIP-graph capture of the evaluation loops.

A flow evaluation is a fixed sequence of kernel launches on one stream (one per layer on the
fused paths, a handful per layer otherwise) with no host-side decisions in between: the kernels
take device pointers and sizes only, never synchronise and never allocate (outputs come from
PyTorch's caching allocator, which records into the graph's private pool).  At small and medium
batch sizes the launch gaps dominate (config C1 at batch 4096: ~20 launches of a few microseconds
of work each), so the whole ``log_prob`` / ``sample_from`` call is captured once into a HIP graph
and replayed with a single launch.

    g = GraphedFlow(model, batch=4096, context_features=16)
    log_q = g.log_prob(x, context)          # copies into the static inputs, replays, returns views
    z, log_q = g.sample_from(eps, context)

Shapes are fixed at capture time.  Parameters the kernels read directly are device pointers, so
in-place weight updates are seen by later replays; the fused layer kernels read PACKED copies of
their conditioner weights, which are refreshed (in place, same address) by ``refresh()`` - call it
after an optimiser step.  A change of parameter STORAGE (``.to()``, ``load_state_dict`` with
assign) needs a new GraphedFlow.  Inference only (no autograd through a graph).
"""
import torch


class GraphedFlow:
    def __init__(self, model, batch, context_features=None, warmup=3):
        p = next(model.parameters())
        dev, dt = p.device, p.dtype
        if dev.type != 'cuda':
            raise ValueError("GraphedFlow needs the model on a HIP device")
        self.model = model
        shape = tuple(model.q0.shape)
        self.x = torch.zeros((batch,) + shape, dtype=dt, device=dev)
        self.eps = torch.zeros((batch,) + shape, dtype=dt, device=dev)
        self.ctx = None if context_features is None else torch.zeros(batch, context_features, dtype=dt, device=dev)
        self._graphs = {}
        self._warmup = warmup

    def _capture(self, name, fn):
        # side stream warm-up (lazy caches: int32 index vectors, packed weights, LDS attributes,
        # the discriminant counter) as the PyTorch graph API requires, then the capture itself
        s = torch.cuda.Stream(device=self.x.device)
        s.wait_stream(torch.cuda.current_stream(self.x.device))
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(self._warmup):
                fn()
        torch.cuda.current_stream(self.x.device).wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(g):
            out = fn()
        self._graphs[name] = (g, out)
        return g, out

    def refresh(self):
        """Re-pack the fused kernels' weight copies after the parameters changed (one eager
        evaluation outside the graph; the packed buffers are rewritten in place)."""
        kw = {} if self.ctx is None else {'context': self.ctx}
        with torch.no_grad():
            self.model.log_prob(self.x, **kw)

    def _ctx_kw(self, context):
        if (context is None) != (self.ctx is None):
            raise ValueError("context must be given exactly when the graph was built with context_features")
        if context is not None:
            self.ctx.copy_(context)
            return {'context': self.ctx}
        return {}

    def log_prob(self, x, context=None):
        kw = self._ctx_kw(context)
        self.x.copy_(x)
        g, out = self._graphs.get('log_prob') or self._capture('log_prob', lambda: self.model.log_prob(self.x, **kw))
        g.replay()
        return out

    def sample_from(self, eps, context=None):
        kw = self._ctx_kw(context)
        self.eps.copy_(eps)
        g, out = self._graphs.get('sample') or self._capture('sample', lambda: self.model.sample_from(self.eps, **kw))
        g.replay()
        return out


class GraphedTrainStep:
    """One optimiser step - ``zero_grad``, a ``NormalizingFlow`` objective, ``backward``, ``optimizer.step`` - captured
    into a HIP graph and replayed with a single launch.

    At the batch sizes the reference's own drivers train with (1024 - 2048 samples, /root/reference/run.py:45-47,
    165-171) a step of the C3 model is ~1500 kernel launches of a few microseconds of work each: the step is bound by
    the host (25 ms per step measured at 16 384 samples, the GPU busy for a third of it).  Every kernel of the training
    path takes device pointers and sizes only, never synchronises and allocates through PyTorch's caching allocator,
    so the whole step records into one graph (the standard whole-network capture recipe of ``torch.cuda.graphs``).

        opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True, fused=True)   # fused: ONE kernel for all
                                                    # parameters (foreach-Adam is ~650 tiny launches on a 450-tensor model)
        step = nf.GraphedTrainStep(model, opt, batch=2048, context_features=16)
        for x, ctx in loader:
            loss = step(x, ctx)              # device scalar of THIS step; read it (``float(loss)``) only when needed

    ``loss``: 'forward_kld' (core.py:30-65) or a callable ``f(model, x, context) -> scalar``.  The optimiser must keep
    its state on the device (``capturable=True`` for Adam / AdamW; SGD is capturable as it is).  Shapes are fixed at
    capture time; the parameters are updated in place, so eager evaluations in between (``model.log_prob``) see the
    current weights (their packed copies refresh on the parameters' version counters)."""

    def __init__(self, model, optimizer, batch, context_features=None, loss='forward_kld', warmup=3):
        p = next(model.parameters())
        dev, dt = p.device, p.dtype
        if dev.type != 'cuda':
            raise ValueError("GraphedTrainStep needs the model on a HIP device")
        for grp in optimizer.param_groups:
            if 'capturable' in grp and not grp['capturable']:
                raise ValueError("the optimiser keeps its step count on the host: construct it with capturable=True")
        self.model, self.optimizer = model, optimizer
        shape = tuple(model.q0.shape)
        self.x = torch.zeros((batch,) + shape, dtype=dt, device=dev)
        self.ctx = None if context_features is None else torch.zeros(batch, context_features, dtype=dt, device=dev)
        if callable(loss):
            self._loss = loss
        elif loss == 'forward_kld':
            self._loss = lambda m, x, c: m.forward_kld(x, context=c) if c is not None else m.forward_kld(x)
        else:
            raise ValueError("loss: 'forward_kld' or a callable (model, x, context) -> scalar")
        self._graph = None
        self._out = None
        self._warmup = warmup

    def _one(self):
        self.optimizer.zero_grad(set_to_none=True)
        out = self._loss(self.model, self.x, self.ctx)
        out.backward()
        self.optimizer.step()
        return out

    def _capture(self):
        s = torch.cuda.Stream(device=self.x.device)
        s.wait_stream(torch.cuda.current_stream(self.x.device))
        with torch.cuda.stream(s):
            for _ in range(self._warmup):          # real steps on the current inputs: lazy state (optimiser moments,
                self._one()                        # packed buffers, workspaces) exists before the capture
        torch.cuda.current_stream(self.x.device).wait_stream(s)
        self._graph = torch.cuda.CUDAGraph()
        self.optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self._graph):
            self._out = self._one()

    def __call__(self, x, context=None):
        if (context is None) != (self.ctx is None):
            raise ValueError("context must be given exactly when the step was built with context_features")
        self.x.copy_(x)
        if context is not None:
            self.ctx.copy_(context)
        if self._graph is None:
            self._capture()                        # note: the warm-up performs ``warmup`` real steps on this first batch
        self._graph.replay()
        return self._out
