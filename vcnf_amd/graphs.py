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
