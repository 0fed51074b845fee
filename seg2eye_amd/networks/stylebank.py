"""StyleBank: the style FCs of every SPADE+Style layer of a generator as ONE GEMM.

Each SPADE_STYLE_Block owns an ApplyStyle FC (normalization.py:144-169 of the reference):
style_i = LeakyReLU(w @ W_i^T + b_i), (N, w_dim) -> (N, 2C_i).  A generator has 21 of them, all fed the
same latent w.  Run one by one on stock torch they cost ~150 launches per step for a few kFLOP each
(GEMV + bias + LeakyReLU, their backward, per-layer gradient accumulation): ~1 ms of a 34 ms step.  The
bank evaluates them together:

    big = LeakyReLU(w @ [W_1; ...; W_21]^T + [b_1; ...; b_21])          (N, S),  S = sum 2C_i

Layer i reads its code as columns [off_i, off_i + 2C_i) of `big` (the modulation kernels take a leading
dimension) and its backward adds d style_i into the same columns of one accumulator `dbig`; the bank's
backward turns dbig into dW, db, dw with two GEMMs.  When Pix2PixModel.create_optimizers has laid the FC
weights (and biases) back to back in the optimizer arena, [W_1; ...] is a zero-copy view and dW / db
are accumulated straight into the gradient arena; otherwise the weights are concatenated (autograd
splits the gradient again) -- same numbers, a few more launches.

state_dict keys, shapes and the per-module parameters are untouched."""
import torch
import torch.nn.functional as F

from .. import ops
from .. import _lib as L

_current = None


def current():
    return _current


class StyleBankFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, Wcat, bcat, holder, gW, gb, want_grad, anchor=None):
        N, K = w.shape
        # the HIP kernels (csrc/style_fc.hip) take the arena layout -- contiguous fp32 stacks -- and the usual small N, w_dim
        ctx.hip = bool(w.is_cuda and Wcat.is_contiguous() and bcat.is_contiguous() and Wcat.dtype == torch.float32
                       and L.lib().s2e_style_fc_supported(N, K))
        if ctx.hip:
            big = torch.empty(N, Wcat.shape[0], dtype=torch.float32, device=w.device)
            L.check(L.lib().s2e_style_fc_fwd(w.data_ptr(), Wcat.data_ptr(), bcat.data_ptr(), big.data_ptr(), N, K, Wcat.shape[0], 0.2,
                                             torch.cuda.current_stream().cuda_stream), 's2e_style_fc_fwd')
        else:
            big = F.leaky_relu(torch.addmm(bcat, w, Wcat.t()), 0.2)
        ctx.set_materialize_grads(False)
        ctx.dst = (gW, gb)
        ctx.dbig = None
        if want_grad:                                   # (grad mode is always off INSIDE a Function.forward)
            ctx.dbig = ops.ZeroPool.take(big.numel(), torch.float32, big.device).view(big.shape)
        holder.append(ctx.dbig)
        ctx.save_for_backward(w, Wcat, big)
        return big

    @staticmethod
    def backward(ctx, gbig):
        w, Wcat, big = ctx.saved_tensors
        gW, gb = ctx.dst
        if ctx.hip and gW is not None and gW.is_contiguous() and gb.is_contiguous():
            N, K = w.shape
            S = Wcat.shape[0]
            want_dw = ctx.needs_input_grad[0]
            dw = torch.empty(N, K, dtype=torch.float32, device=w.device) if want_dw else None
            wsb = L.lib().s2e_style_fc_bwd_workspace_bytes(N, K, S) if want_dw else 0
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=w.device) if wsb else None
            gbc = None if gbig is None else gbig.float().contiguous()
            L.check(L.lib().s2e_style_fc_bwd(ctx.dbig.data_ptr(), None if gbc is None else gbc.data_ptr(), big.data_ptr(), w.data_ptr(),
                                             Wcat.data_ptr(), gW.data_ptr(), gb.data_ptr(), None if dw is None else dw.data_ptr(),
                                             None if ws is None else ws.data_ptr(), wsb, N, K, S, 0.2,
                                             torch.cuda.current_stream().cuda_stream), 's2e_style_fc_bwd')
            return dw, None, None, None, None, None, None, None
        d = ctx.dbig if gbig is None else ctx.dbig + gbig          # consumers accumulate into dbig directly
        dpre = torch.where(big > 0, d, 0.2 * d)                     # LeakyReLU': big > 0 <=> pre > 0
        dw = dpre @ Wcat if ctx.needs_input_grad[0] else None
        if gW is not None:
            gW.addmm_(dpre.t(), w)
            gb.add_(dpre.sum(0))
            return dw, None, None, None, None, None, None, None
        return dw, dpre.t() @ w, dpre.sum(0), None, None, None, None, None


class StyleBank:
    def __init__(self, root):
        from .normalization import SPADE_STYLE_Block
        self.fcs = [m.adain.linear for m in root.modules() if isinstance(m, SPADE_STYLE_Block)]
        self.offsets, off = {}, 0
        for f in self.fcs:
            self.offsets[id(f)] = off
            off += f.weight.shape[0]
        self.S = off
        self.usable = bool(self.fcs) and all(f.bias is not None and f.w_lrmul == 1.0 and f.b_lrmul == 1.0 for f in self.fcs)

    @staticmethod
    def _chain(ts):
        return all(ops._adjacent(a, b) for a, b in zip(ts[:-1], ts[1:]))

    def run(self, w):
        """-> (big, dbig): the (N,S) style matrix and its gradient accumulator (None without autograd)."""
        Ws, bs = [f.weight for f in self.fcs], [f.bias for f in self.fcs]
        k = Ws[0].shape[1]
        gW = gb = None
        # the arena views are memoised on the addresses of the parameters and their gradients (84 adjacency checks per forward
        # otherwise); they alias the arenas and stay valid while the parameters stay where they are
        key = tuple(p.data_ptr() for p in Ws + bs) + tuple(0 if p.grad is None else p.grad.data_ptr() for p in Ws + bs)
        memo = self.__dict__.get('_views')
        if memo is not None and memo[0] == key:
            Wcat, bcat, gW, gb = memo[1]
        elif self._chain(Ws) and self._chain(bs):
            Wcat, bcat = ops._span2(Ws[0], (self.S, k)), ops._span2(bs[0], (self.S,))
            gWs, gbs = [ops._grad_dst(p) for p in Ws], [ops._grad_dst(p) for p in bs]
            if all(g is not None for g in gWs + gbs) and self._chain(gWs) and self._chain(gbs):
                gW, gb = ops._span2(gWs[0], (self.S, k)), ops._span2(gbs[0], (self.S,))
                self.__dict__['_views'] = (key, (Wcat, bcat, gW, gb))
            else:                                   # weights adjacent but no gradient arena: let autograd split
                Wcat, bcat = torch.cat(Ws, 0), torch.cat(bs, 0)
        else:
            Wcat, bcat = torch.cat(Ws, 0), torch.cat(bs, 0)
        holder = []
        # In the arena path Wcat / bcat are detached views and dW / db are written by the backward itself, so autograd sees no
        # input that needs a gradient when `w` does not (a style code passed in directly -- generate_fake_from_stylecode --, a
        # frozen or detached netE): the node would be pruned and all 21 FC gradients silently dropped.  The first FC weight
        # rides along as an `anchor` input (its gradient slot returns None) to keep the node in the graph.
        anchor = Ws[0] if (gW is not None and torch.is_grad_enabled() and Ws[0].requires_grad) else None
        big = StyleBankFn.apply(w.float().contiguous(), Wcat, bcat, holder, gW, gb, torch.is_grad_enabled(), anchor)
        return big, holder[0]


class scope:
    """`with stylebank.scope(generator, w):` around the generator's blocks."""

    def __init__(self, root, w):
        bank = root.__dict__.get('_style_bank')
        if bank is None:
            bank = StyleBank(root)
            root.__dict__['_style_bank'] = bank
        self.bank, self.w = bank, w

    def __enter__(self):
        global _current
        self.prev = _current
        if self.bank.usable:
            big, dbig = self.bank.run(self.w)
            _current = (self.bank.offsets, big, dbig)
        else:
            _current = None
        return self

    def __exit__(self, *exc):
        global _current
        _current = self.prev
        return False
