"""Training under FROZEN BatchNorm on the library's row kernels.

The reference's ImageNet recipe is dense by construction (BatchNorm in batch-statistics mode, `models/laud_resnet.py:115-133`): nothing
to accelerate.  Its DETECTION fine-tuning freezes the BatchNorm STATISTICS (`norm_eval=True`, `mmdetection-2.21.0/mmdet/models/backbones/
lad_mmdet_resnet.py:753-758`) while the affine parameters keep training (`norm_cfg=dict(type='BN', requires_grad=True)`); the configs it
ships are channel-2222 and layer (`configs/retinanet/scale_backbone_lr/retinanet_ladmmdet_r101_fpn_1x_coco_r101_channel_2222_0x6_lrmult0x2.py:8-17`,
`..._layer_0x5_lrmult0x2.py:8-17`).  Under frozen statistics a block is

    out = relu(identity(x) + m3 * bn3(conv3(relu(bn2(c . conv2(relu(bn1(c . conv1(x)))))))))        (laud_resnet.py:104-147)

with a {0,1} pixel mask m3 (spatial / layer modes) or a {0,1} channel mask c applied BEFORE bn1 / bn2 (channel mode), both sampled with
F.gumbel_softmax(hard=True) in training (`models/utils.py:56-58,124,162`).  This module runs the three convolutions -- forward AND backward --
on the library's kernels:

  pixel masks (spatial, layer; any stride, with or without a projection shortcut): the packed rows of the kept pixels
      forward   x --conv1 (rows of the dilated list)--> h1 --3x3 through the neighbour table--> h2 --conv3, scattered--> branch
      backward  the same kernels with gather and scatter swapped and transposed weights; the 3x3 goes through the TRANSPOSED neighbour
                table (for every h1 row and tap the packed output row whose window reads it; stride-aware).  A skipped image / dropped
                pixel costs nothing in either direction.
  channel masks (any stride): the row kernels over ALL pixels with the channel mask and the constants of the channel algebra as epilogue
      terms (a masked channel is the constant relu(shift); the library's "dense channel execution").  Correct for every
      gradient, but dense in the channels: the masks save no FLOPs here (stated, not hidden).
  Weight gradients are library GEMMs over the packed rows; the gradients of BatchNorm's affine parameters come out of the folded
  (scale, shift) pairs, which stay differentiable functions of (weight, bias); the straight-through terms of the hard masks need the branch at
  DROPPED units too -- computed by the library's own dense execution (that term is dense in the reference as well).
  The residual add, the final ReLU, the projection shortcut, the maskers' tiny heads, the static stem and the classifier are plain autograd
  ops.

Entry points: `sparse_block_train(block, x, mask)` (one block, the mask an input), `block_train(block, state, temperature)` (the reference's
block signature in training mode: samples its own masks), `train_forward(model, x, temperature)` (a whole LAUD-ResNet -> the reference's
7-tuple; a `LAD_MMDet_ResNet` -> its (outs, additional, model_configs)), `prepare_for_training(model)` (train mode with BatchNorm statistics
frozen).  Checked against the oracle's autograd -- blocks on the reference-generated block fixtures, whole models on `det_tiny.pt` /
`full_tiny.pt` with identical Gumbel noise (tests/test_hip_training.py, plain 1e-3).  Not built: dyn_mode 'both', mask groups > 1, grouped /
dilated conv2, BatchNorm in batch-statistics mode."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import ops
from ._lib import LdnError


# ------------------------------------------------------------------------------------------------------------------ index helpers
def transposed_neighbour_table(ix, B, H, W, stride=1, Ho=None, Wo=None):
    """nbrT [cap1, 9]: for every packed h1 row r (an INPUT pixel (iy, ix) of the dilated list) and tap t = (dy, dx), the packed OUTPUT row whose
    3x3 window reads r through tap t: output pixel (oy, ox) with stride * oy + dy - 1 == iy (likewise x), -1 if there is none / it is not active.
    The adjoint of ix.nbr: d h1[r] = sum_t d u2[nbrT[r, t]] . W2[:, t, :].  H, W = the input map; Ho, Wo = the output map (stride 1: the same)."""
    Ho = H if Ho is None else Ho
    Wo = W if Wo is None else Wo
    dev = ix.idx1.device
    cap1 = ix.cap1
    pix = ix.idx1.long().clamp(min=0)                       # flat input pixel b * H * W + iy * W + ix of every list entry (garbage past the count)
    b = pix // (H * W)
    rem = pix - b * (H * W)
    iy, ixx = rem // W, rem - (rem // W) * W
    pos3 = ix.pos3.view(-1).long()
    taps = torch.arange(9, device=dev)
    dy, dx = taps // 3 - 1, taps % 3 - 1                     # all nine taps at once: [cap1, 9] tensors, a dozen launches instead of a hundred
    ny, nx = iy[:, None] - dy[None, :], ixx[:, None] - dx[None, :]      # = stride * oy, stride * ox
    oy, ox = torch.div(ny, stride, rounding_mode="floor"), torch.div(nx, stride, rounding_mode="floor")
    ok = (ny >= 0) & (nx >= 0) & (oy * stride == ny) & (ox * stride == nx) & (oy < Ho) & (ox < Wo)
    q = (b[:, None] * (Ho * Wo) + oy.clamp(0, Ho - 1) * Wo + ox.clamp(0, Wo - 1)).clamp(0, pos3.numel() - 1)
    out = torch.where(ok, pos3[q], torch.full_like(q, -1)).to(torch.int32)
    valid = torch.arange(cap1, device=dev) < ix.cnt[1]
    return torch.where(valid[:, None], out, torch.full_like(out, -1)).contiguous()


_DENSE_IX = {}


def _dense_lists(B, Ho, Wo, stride, dev):
    """Index lists of an all-active batch, cached per shape (Bottleneck._dense_ix)."""
    key = (B, Ho, Wo, stride, str(dev))
    if key not in _DENSE_IX:
        if len(_DENSE_IX) > 64:
            _DENSE_IX.clear()
        _DENSE_IX[key] = ops.mask_to_index(torch.ones(B, 1, 1, device=dev), Ho, Wo, stride)
    return _DENSE_IX[key]


def _rows_valid(n_cap, count, dev):
    """[n_cap, 1] float mask of the rows in front of a device-side count (no host read)."""
    return (torch.arange(n_cap, device=dev) < count).to(torch.float32).unsqueeze(1)


def _weight_grad_3x3(du2, h1, nbr, cap1, count3=None):
    """d W2 [n, k, 3, 3] = sum over packed output rows of du2[row, n] * h1[nbr[row, tap], k] (a missing neighbour is a zero row).  Rows past the
    device-side count carry du2 == 0 and are pointed at the zero row (their table entries are uninitialised): no host read is needed."""
    W = du2.shape[1]
    h1z = torch.cat((h1, torch.zeros(1, h1.shape[1], device=h1.device)))
    nb = nbr.view(-1, 9)
    if count3 is not None:
        nb = torch.where((torch.arange(nb.shape[0], device=nb.device) < count3).unsqueeze(1), nb, torch.full_like(nb, -1))
    nb = torch.where(nb < cap1, nb, torch.full_like(nb, -1)).long()
    nb = torch.where(nb >= 0, nb, torch.full_like(nb, cap1))
    cols = h1z[nb.reshape(-1)].view(nb.shape[0], 9 * h1.shape[1])          # [rows, 9 K]: the nine taps' rows side by side -- ONE GEMM
    return (du2.t() @ cols).view(W, 9, h1.shape[1]).permute(0, 2, 1).reshape(W, h1.shape[1], 3, 3)


# ------------------------------------------------------------------------------------------------------------------ pixel masks
class _PixelBranchFn(torch.autograd.Function):
    """branch = m3 * (s3 * conv3(relu(s2 * conv2(relu(s1 * conv1(x) + t1)) + t2)) + t3)  on the packed rows of the kept pixels (NCHW out, zeros at
    the dropped pixels).  Differentiable in x, the three weights, the six folded BatchNorm vectors and the mask."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, m3, s1, t1, s2, t2, s3, t3, stride):
        B, Cin, Hi, Wi = x.shape
        Ho, Wo = m3.shape[2], m3.shape[3]
        W, cout = w1.shape[0], w3.shape[0]
        dev = x.device
        xn = ops.as_nhwc(x.detach())
        x2d = xn.reshape(B * Hi * Wi, Cin)
        ix = ops.mask_to_index(m3.detach().reshape(B, Ho, Wo).contiguous().float(), Ho, Wo, stride)
        s1, t1, s2, t2, s3, t3 = (v.detach().float().contiguous() for v in (s1, t1, s2, t2, s3, t3))
        w1r = w1.detach().reshape(W, 1, Cin).float().contiguous()
        w2r = w2.detach().permute(0, 2, 3, 1).reshape(W, 9, W).float().contiguous()
        w3r = w3.detach().reshape(cout, W).float()
        w3s = (w3r * s3.view(-1, 1)).reshape(cout, 1, W).contiguous()
        h1 = torch.zeros(ix.cap1, W, device=dev)
        ops.conv_rows(x2d, w1r, s1, t1, h1, a_rows=ix.idx1, taps=1, m_count=ix.cnt[1:2], m_cap=ix.cap1)
        h2 = torch.zeros(ix.cap3, W, device=dev)
        ops.conv_rows(h1, w2r, s2, t2, h2, a_rows=ix.nbr, taps=9, m_count=ix.cnt[0:1], m_cap=ix.cap3)
        br = torch.zeros(B * Ho * Wo, cout, device=dev)
        ops.conv_rows(h2, w3s, None, t3, br, taps=1, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=0, out_rows=ix.idx3)
        ctx.save_for_backward(x2d, h1, h2, br, w1r, w2r, w3r, s1, t1, s2, t2, s3, t3, m3.detach().float())
        ctx.ix, ctx.shape, ctx.stride = ix, (B, Cin, Hi, Wi, Ho, Wo, W, cout), stride
        return ops.from_nhwc(br.view(B, Ho, Wo, cout))

    @staticmethod
    def backward(ctx, g):
        x2d, h1, h2, br, w1r, w2r, w3r, s1, t1, s2, t2, s3, t3, m3d = ctx.saved_tensors
        ix, stride = ctx.ix, ctx.stride
        B, Cin, Hi, Wi, Ho, Wo, W, cout = ctx.shape
        dev = g.device
        need = ctx.needs_input_grad
        go = ops.as_nhwc(g.contiguous()).reshape(B * Ho * Wo, cout)
        zW, zC = torch.zeros(W, device=dev), torch.zeros(Cin, device=dev)
        v3, v1 = _rows_valid(ix.cap3, ix.cnt[0], dev), _rows_valid(ix.cap1, ix.cnt[1], dev)
        # conv3^T on the active rows
        g3 = ops.gather_rows(go, ix.idx3, count=ix.cnt[0:1], cap=ix.cap3) * v3          # d L / d (s3 y3 + t3) at the kept pixels
        w3s = w3r * s3.view(-1, 1)
        dh2 = torch.zeros(ix.cap3, W, device=dev)
        ops.conv_rows(g3, w3s.t().reshape(W, 1, cout).contiguous(), None, zW, dh2, taps=1, m_count=ix.cnt[0:1], m_cap=ix.cap3, relu=0)
        dz2 = dh2 * (h2 > 0) * v3                         # through ReLU: d L / d (s2 y2 + t2)
        du2 = dz2 * s2
        # conv2^T: the 3x3 through the transposed neighbour table
        nbrT = transposed_neighbour_table(ix, B, Hi, Wi, stride, Ho, Wo)
        dh1 = torch.zeros(ix.cap1, W, device=dev)
        ops.conv_rows(du2, w2r.permute(2, 1, 0).contiguous(), None, zW, dh1, a_rows=nbrT, taps=9, m_count=ix.cnt[1:2], m_cap=ix.cap1, relu=0)
        dz1 = dh1 * (h1 > 0) * v1
        du1 = dz1 * s1
        grad_x = None
        if need[0]:
            gx = torch.zeros(B * Hi * Wi, Cin, device=dev)
            ops.conv_rows(du1, w1r.reshape(W, Cin).t().reshape(Cin, 1, W).contiguous(), None, zC, gx, taps=1, m_count=ix.cnt[1:2], m_cap=ix.cap1,
                          relu=0, out_rows=ix.idx1, residual2d=gx)
            grad_x = ops.from_nhwc(gx.view(B, Hi, Wi, Cin))
        # weight gradients: library GEMMs over the packed rows (rows past the counts are zero in g3 / du2 / du1)
        gw1 = gw2 = gw3 = None
        if need[3]:
            gw3 = ((g3.t() @ h2) * s3.view(-1, 1)).reshape(cout, W, 1, 1)
        if need[2]:
            gw2 = _weight_grad_3x3(du2, h1, ix.nbr, ix.cap1, ix.cnt[0])
        if need[1]:
            rows1 = torch.where(v1.squeeze(1) > 0, ix.idx1.long(), torch.zeros_like(ix.idx1, dtype=torch.long)).clamp(0, x2d.shape[0] - 1)
            gw1 = (du1.t() @ x2d[rows1]).reshape(W, Cin, 1, 1)      # (list entries past the count are uninitialised: du1 is zero there)
        # folded BatchNorm vectors: z = s y + t  =>  d t = sum d z,  d s = sum d z * y,  y = (z - t) / s wherever d z != 0 (there z = the stored ReLU output)
        safe = lambda s: torch.where(s == 0, torch.ones_like(s), s)
        gs1 = (dz1 * (h1 - t1)).sum(0) / safe(s1) if need[5] else None
        gt1 = dz1.sum(0) if need[6] else None
        gs2 = (dz2 * (h2 - t2)).sum(0) / safe(s2) if need[7] else None
        gt2 = dz2.sum(0) if need[8] else None
        y3s = ops.gather_rows(br, ix.idx3, count=ix.cnt[0:1], cap=ix.cap3) - t3       # = s3 * y3 at the kept pixels
        gs3 = (g3 * y3s).sum(0) / safe(s3) if need[9] else None
        gt3 = g3.sum(0) if need[10] else None
        gm = None
        if need[4]:
            # straight-through term of the hard mask: d L / d m3[p] = sum_c g[p, c] * (s3 conv3(..) + t3)[p, c] needs the branch at the DROPPED pixels
            # too (that term is dense in the reference as well).  The kept pixels' branch is `br`; the dropped ones' comes from the same three
            # launches over the COMPLEMENT's lists -- together one dense execution of the block, of which the forward already paid the kept part.
            cix = ops.mask_to_index((1.0 - m3d).reshape(B, Ho, Wo).contiguous(), Ho, Wo, stride)
            d1 = torch.zeros(cix.cap1, W, device=dev)
            ops.conv_rows(x2d, w1r, s1, t1, d1, a_rows=cix.idx1, taps=1, m_count=cix.cnt[1:2], m_cap=cix.cap1)
            d2 = torch.zeros(cix.cap3, W, device=dev)
            ops.conv_rows(d1, w2r, s2, t2, d2, a_rows=cix.nbr, taps=9, m_count=cix.cnt[0:1], m_cap=cix.cap3)
            full = br.clone()
            ops.conv_rows(d2, w3s.reshape(cout, 1, W).contiguous(), None, t3, full, taps=1, m_count=cix.cnt[0:1], m_cap=cix.cap3, relu=0, out_rows=cix.idx3)
            gm = (go * full).sum(dim=1).view(B, 1, Ho, Wo)
        return grad_x, gw1, gw2, gw3, gm, gs1, gt1, gs2, gt2, gs3, gt3, None


# ------------------------------------------------------------------------------------------------------------------ channel masks
def _channel_constants(w2, w3, s2, t2, t1, s3, t3):
    """The constants of the channel algebra for the CURRENT weights: c1 = relu(t1), c2 = relu(t2), the 16 border classes of
    t2 + s2 * (W2 (*) c1), and t3 + s3 * (W3 c2)."""
    W = w2.shape[0]
    c1, c2 = torch.relu(t1), torch.relu(t2)
    wc = torch.einsum("okyx,k->oyx", w2, c1)
    tab = torch.empty(16, W, device=w2.device)
    for cls in range(16):
        rb, cb = cls // 4, cls % 4
        ys = [ky for ky in range(3) if not ((ky == 0 and rb & 1) or (ky == 2 and rb & 2))]
        xs = [kx for kx in range(3) if not ((kx == 0 and cb & 1) or (kx == 2 and cb & 2))]
        tab[cls] = t2 + s2 * wc[:, ys][:, :, xs].sum(dim=(1, 2))
    t3c = t3 + s3 * (w3.reshape(w3.shape[0], W) @ c2)
    return c1.contiguous(), c2.contiguous(), tab.contiguous(), t3c.contiguous()


class _ChannelBranchFn(torch.autograd.Function):
    """branch = s3 * conv3(relu(s2 * (c . conv2(relu(s1 * (c . conv1(x)) + t1))) + t2)) + t3  with a {0,1} channel mask c [B, W] applied before bn1 / bn2
    (laud_resnet.py:116-118,124-126), on the row kernels over all pixels: u = relu(bn(.)) - relu(shift) is stored, zeroed on the masked channels
    of each image (the library's dense channel execution, Bottleneck._run_channel_dense)."""

    @staticmethod
    def forward(ctx, x, w1, w2, w3, chm, s1, t1, s2, t2, s3, t3, stride):
        B, Cin, Hi, Wi = x.shape
        Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
        W, cout = w1.shape[0], w3.shape[0]
        dev = x.device
        xn = ops.as_nhwc(x.detach())
        x2d = xn.reshape(B * Hi * Wi, Cin)
        s1, t1, s2, t2, s3, t3 = (v.detach().float().contiguous() for v in (s1, t1, s2, t2, s3, t3))
        w2f, w3f = w2.detach().float(), w3.detach().float()
        c1, c2, tab, t3c = _channel_constants(w2f, w3f, s2, t2, t1, s3, t3)
        w1r = w1.detach().reshape(W, 1, Cin).float().contiguous()
        w2r = w2f.permute(0, 2, 3, 1).reshape(W, 9, W).contiguous()
        w3r = w3f.reshape(cout, W)
        w3s = (w3r * s3.view(-1, 1)).reshape(cout, 1, W).contiguous()
        chm2d = chm.detach().float().reshape(B, W).contiguous()
        ix = _dense_lists(B, Ho, Wo, stride, dev)
        fused = ops.dense_kernel_ok() and Cin % 32 == 0 and W % 32 == 0
        h1 = torch.empty(ix.cap1, W, device=dev)
        if fused:
            ops.conv_rows(x2d, w1r, s1, t1, h1, taps=1, m_cap=ix.cap1, relu=1, post_sub=c1, chan_mask=chm2d, rows_per_image=Hi * Wi)
        else:
            ops.conv_packed(x2d, w1r, s1, t1, h1, taps=1, m_cap=ix.cap1, post_sub=c1, relu=1)
            h1.view(B, -1, W).mul_(chm2d.view(B, 1, W))
        h2 = torch.empty(ix.cap3, W, device=dev)
        if fused and 9 in ops.DENSE_TAPS and ops.DENSE_CHANNEL_3X3:
            ops.conv_rows(h1, w2r, s2, tab, h2, a_rows=ix.nbr, taps=9, m_cap=ix.cap3, pix_map=ix.idx3, geom=(Hi, Wi, Ho, Wo, stride), post_sub=c2,
                          relu=1, chan_mask=chm2d, rows_per_image=Ho * Wo)
        else:
            ops.conv_packed(h1, w2r, s2, tab, h2, a_map=ix.nbr, taps=9, m_cap=ix.cap3, pix_map=ix.idx3, geom=(Hi, Wi, Ho, Wo, stride), post_sub=c2, relu=1)
            h2.view(B, -1, W).mul_(chm2d.view(B, 1, W))
        br = torch.empty(B * Ho * Wo, cout, device=dev)
        ops.conv_rows(h2, w3s, None, t3c, br, taps=1, m_cap=ix.cap3, relu=0)
        ctx.save_for_backward(x2d, h1, h2, br, w1r, w2r, w3r, s1, t1, s2, t2, s3, t3, chm2d, c1, c2, tab)
        ctx.ix, ctx.shape, ctx.stride = ix, (B, Cin, Hi, Wi, Ho, Wo, W, cout), stride
        return ops.from_nhwc(br.view(B, Ho, Wo, cout))

    @staticmethod
    def backward(ctx, g):
        x2d, u1, u2, br, w1r, w2r, w3r, s1, t1, s2, t2, s3, t3, chm2d, c1, c2, tab = ctx.saved_tensors
        ix, stride = ctx.ix, ctx.stride
        B, Cin, Hi, Wi, Ho, Wo, W, cout = ctx.shape
        dev = g.device
        need = ctx.needs_input_grad
        go = ops.as_nhwc(g.contiguous()).reshape(B * Ho * Wo, cout)
        zW, zC = torch.zeros(W, device=dev), torch.zeros(Cin, device=dev)
        cm3 = chm2d.view(B, 1, W)
        per_img = lambda t2d: t2d.view(B, -1, W)
        # forward values with the masks applied: h = u + c at every channel (a masked channel is the constant c), z > 0 <=> h > 0
        h1f, h2f = u1 + c1, u2 + c2
        on1 = (per_img(h1f) > 0).float()
        on2 = (per_img(h2f) > 0).float()
        w3s = w3r * s3.view(-1, 1)
        dh2 = torch.empty(ix.cap3, W, device=dev)                     # d L / d h2 at EVERY channel
        ops.conv_rows(go, w3s.t().reshape(W, 1, cout).contiguous(), None, zW, dh2, taps=1, m_cap=ix.cap3, relu=0)
        dz2_all = per_img(dh2) * on2                                  # d L / d z2 (z2 = s2 * (c . y2) + t2)
        dz2 = (dz2_all * cm3).reshape(-1, W)
        du2 = dz2 * s2                                                # d L / d y2 on the active channels
        nbrT = transposed_neighbour_table(ix, B, Hi, Wi, stride, Ho, Wo)
        dh1 = torch.empty(ix.cap1, W, device=dev)                     # d L / d h1 at every channel
        ops.conv_rows(du2, w2r.permute(2, 1, 0).contiguous(), None, zW, dh1, a_rows=nbrT, taps=9, m_cap=ix.cap1, relu=0)
        dz1_all = per_img(dh1) * on1
        dz1 = (dz1_all * cm3).reshape(-1, W)
        du1 = dz1 * s1
        grad_x = None
        if need[0]:
            gx = torch.zeros(B * Hi * Wi, Cin, device=dev)
            ops.conv_rows(du1, w1r.reshape(W, Cin).t().reshape(Cin, 1, W).contiguous(), None, zC, gx, taps=1, m_cap=ix.cap1, relu=0,
                          out_rows=ix.idx1, residual2d=gx)
            grad_x = ops.from_nhwc(gx.view(B, Hi, Wi, Cin))
        gw1 = gw2 = gw3 = None
        if need[3]:     # conv3 sees h2 = u2 + c2 at every channel (the constants of the masked ones included)
            gw3 = ((go.t() @ h2f) * s3.view(-1, 1)).reshape(cout, W, 1, 1)
        if need[2]:     # conv2 sees h1 = u1 + c1 inside the map, zeros in the padding ring
            gw2 = _weight_grad_3x3(du2, h1f, ix.nbr, ix.cap1)
        if need[1]:
            gw1 = (du1.t() @ x2d[ix.idx1.long().clamp(0, x2d.shape[0] - 1)]).reshape(W, Cin, 1, 1)
        safe = lambda s: torch.where(s == 0, torch.ones_like(s), s)
        # z = s (c . y) + t: d t sums d z over EVERY channel's pixels (a masked channel's z = t still feeds the ReLU); d s only sees active channels
        gs1 = (dz1 * (h1f - t1)).sum(0) / safe(s1) if need[5] else None
        gt1 = dz1_all.reshape(-1, W).sum(0) if need[6] else None
        gs2 = (dz2 * (h2f - t2)).sum(0) / safe(s2) if need[7] else None
        gt2 = dz2_all.reshape(-1, W).sum(0) if need[8] else None
        gs3 = (go * (br - t3)).sum(0) / safe(s3) if need[9] else None
        gt3 = go.sum(0) if need[10] else None
        gc = None
        if need[4]:
            # straight-through term: d L / d c[b, k] = sum_p d L / d (c . y)[b, k, p] * y[b, k, p] for both masked products, y = the UNMASKED conv
            # output -- needed at the masked channels too: the library's dense execution without the mask (z = s y + t, no ReLU)
            r1 = torch.empty(ix.cap1, W, device=dev)
            ops.conv_rows(x2d, w1r, s1, t1, r1, a_rows=ix.idx1, taps=1, m_cap=ix.cap1, relu=0)
            y1 = (r1 - t1) / safe(s1)
            r2 = torch.empty(ix.cap3, W, device=dev)
            # (conv2 of h1 = u1 + c1: the constants' share is the border-class table)
            if ops.dense_kernel_ok() and W % 32 == 0 and 9 in ops.DENSE_TAPS and ops.DENSE_CHANNEL_3X3:
                ops.conv_rows(u1, w2r, s2, tab, r2, a_rows=ix.nbr, taps=9, m_cap=ix.cap3, pix_map=ix.idx3, geom=(Hi, Wi, Ho, Wo, stride), relu=0)
            else:
                ops.conv_packed(u1, w2r, s2, tab, r2, a_map=ix.nbr, taps=9, m_cap=ix.cap3, pix_map=ix.idx3, geom=(Hi, Wi, Ho, Wo, stride), relu=0)
            y2 = (r2 - t2) / safe(s2)
            gc = (dz1_all * s1 * per_img(y1)).sum(1) + (dz2_all * s2 * per_img(y2)).sum(1)          # [B, W]
        return grad_x, gw1, gw2, gw3, gc, gs1, gt1, gs2, gt2, gs3, gt3, None


# ------------------------------------------------------------------------------------------------------------------ blocks
def _fold_live(bn):
    """(scale, shift) of a BatchNorm with FROZEN statistics as differentiable functions of its affine parameters."""
    s = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
    return s, bn.bias - bn.running_mean * s


def _check_block(block, x):
    if block.conv2.groups != 1 or block.conv2.dilation[0] != 1:
        raise LdnError("training: grouped / dilated conv2 is not built")
    if not x.is_cuda:
        raise LdnError("laudnet_amd ops need tensors on a HIP device (cuda:N); there is no CPU path")
    for bn in (block.bn1, block.bn2, block.bn3):
        if bn.training:
            raise LdnError("training: BatchNorm must run on its frozen statistics (norm_eval; call prepare_for_training(model)) -- the "
                           "batch-statistics recipe is dense by construction and not built")


def _identity(block, x):
    return x if block.downsample is None else block.downsample(x)      # (conv 1x1 stride s + BatchNorm on frozen statistics: plain autograd)


def sparse_block_train(block, x, mask):
    """Differentiable forward of ONE Bottleneck under frozen BatchNorm statistics with its hard mask as an input.
    dyn_mode 'spatial' / 'layer': mask [B, 1, S, S] {0,1};  dyn_mode 'channel': mask [B, G] {0,1}.  The mask may require grad (the hard Gumbel
    sample of the masker's logits): it receives the straight-through term.  Returns the block's output."""
    _check_block(block, x)
    bn = _fold_live(block.bn1) + _fold_live(block.bn2) + _fold_live(block.bn3)
    if block.dyn_mode in ("spatial", "layer"):
        if block.masker_spatial.mask_channel_group != 1 or mask.dim() != 4 or mask.shape[1] != 1:
            raise LdnError("training: one spatial mask group ([B, 1, S, S])")
        Hi, Wi = x.shape[2], x.shape[3]
        Ho, Wo = (Hi - 1) // block.stride + 1, (Wi - 1) // block.stride + 1
        m3 = F.interpolate(mask, size=(Ho, Wo), mode="nearest") if (mask.shape[2], mask.shape[3]) != (Ho, Wo) else mask     # laud_resnet.py:106
        branch = _PixelBranchFn.apply(x, block.conv1.weight, block.conv2.weight, block.conv3.weight, m3, *bn, block.stride)
    elif block.dyn_mode == "channel":
        W, gran = block.width, block.channel_dyn_granularity
        if mask.dim() != 2 or mask.shape[1] * gran != W:
            raise LdnError("training: channel mask must be [B, G] with G * granularity == width")
        chm = mask.unsqueeze(2).expand(-1, -1, gran).reshape(mask.shape[0], W)            # group j owns channels [j * gran, (j + 1) * gran)
        branch = _ChannelBranchFn.apply(x, block.conv1.weight, block.conv2.weight, block.conv3.weight, chm, *bn, block.stride)
    else:
        raise LdnError("training: dyn_mode 'spatial', 'layer' and 'channel' (the reference's detection configs); 'both' is not built")
    return F.relu(branch + _identity(block, x))


def _gumbel_hard(logits2, temperature):
    """models/utils.py:57,124,162: the hard Gumbel-softmax sample over the (keep, drop) pair, keep component."""
    return F.gumbel_softmax(logits2, dim=1, tau=temperature, hard=True)[:, 0]


def sample_spatial_mask(block, x, temperature):
    """Masker_spatial.forward in training mode (models/utils.py:47-58) as plain autograd ops: -> (mask [B, g, S, S], sparsity, flops)."""
    mk = block.masker_spatial
    ms = mk.mask_size
    pooled = F.adaptive_avg_pool2d(x, ms) if ms < x.shape[2] else x
    flops = pooled.shape[1] * pooled.shape[2] * pooled.shape[3]
    lg = mk.conv(pooled)
    flops += mk.conv_flops_pp * lg.shape[2] * lg.shape[3]
    b, c, h, w = lg.shape
    mask = _gumbel_hard(lg.view(b, 2, c // 2, h, w), temperature)
    return mask, mask.mean(), flops


def sample_channel_mask(block, x, temperature):
    """Masker_channel_MLP / _conv_linear .forward in training mode (models/utils.py:113-127,150-165): -> (mask [B, G], sparsity, flops)."""
    mk = block.masker_channel
    b, c, h, w = x.shape
    if hasattr(mk, "linear"):                       # conv_linear: 1x1 conv + BatchNorm (frozen statistics) + ReLU on the map, GAP, Linear
        y = mk.conv(x)
        flops = y.shape[1] * y.shape[2] * y.shape[3] + mk.masker_flops
        lg = mk.linear(F.adaptive_avg_pool2d(y, (1, 1)).view(b, -1))
    else:
        flops = c * h * w + mk.conv_flops
        lg = mk.conv(F.adaptive_avg_pool2d(x, (1, 1)).view(b, c))
    mask = _gumbel_hard(lg.view(b, 2, lg.shape[1] // 2), temperature)
    return mask, mask.mean(), flops


def block_train(block, state, temperature=1.0):
    """Bottleneck.forward of the reference in TRAINING mode (laud_resnet.py:88-165) under frozen BatchNorm statistics: samples the block's hard
    masks from its maskers' logits (forced_*_mask is honoured), runs the convolutions on the row kernels and keeps the reference's bookkeeping
    (sparsity lists, FLOPs ratio, running FLOPs) differentiable where the reference's is (channel sparsity, conv3's spatial sparsity)."""
    x, s3l, s2l, s1l, csl, percl, flops = state
    dev = x.device
    one = lambda: torch.tensor(1.0, device=dev)
    c_flops = s_flops = 0
    Hi, Wi = x.shape[2], x.shape[3]
    Ho, Wo = (Hi - 1) // block.stride + 1, (Wi - 1) // block.stride + 1
    s1 = s2 = s3 = cs = None
    if block.dyn_mode == "channel":
        if block.forced_channel_mask is not None:
            cmask = block.forced_channel_mask.to(x.dtype)
            cs, c_flops = cmask.mean(), block.masker_channel.flops_for(x)
        else:
            cmask, cs, c_flops = sample_channel_mask(block, x, temperature)
        out = sparse_block_train(block, x, cmask)
        s1, s2, s3 = one(), one(), one()
        block.last_channel_mask = cmask.detach()
    elif block.dyn_mode in ("spatial", "layer"):
        if block.forced_spatial_mask is not None:
            m = block.forced_spatial_mask.to(x.dtype)
            s3, s_flops = m.mean(), block.masker_spatial.flops_for(x)
        else:
            m, s3, s_flops = sample_spatial_mask(block, x, temperature)
        out = sparse_block_train(block, x, m)
        # the dilated masks' means (ExpandMask, laud_resnet.py:107-110: thresholded, no gradient) come out of the list build
        m3 = F.interpolate(m.detach(), size=(Ho, Wo), mode="nearest")
        st = ops.mask_to_index(m3.reshape(x.shape[0], Ho, Wo).contiguous().float(), Ho, Wo, block.stride).stats
        s2, s1 = st[1], st[2]
        cs = one()
        block.last_spatial_mask = m.detach()
    else:
        raise LdnError("training: dyn_mode 'spatial', 'layer' and 'channel'")
    W, cin, cout = block.width, block.conv1.in_channels, block.conv3.out_channels
    macs = (cin * W, W * W * 9, W * cout)
    dense = c_flops + s_flops
    sparse = c_flops + s_flops
    dense += macs[0] * Hi * Wi
    sparse = sparse + macs[0] * Hi * Wi * cs * s1
    dense += macs[1] * Ho * Wo
    sparse = sparse + macs[1] * Ho * Wo * cs ** 2 * s2
    dense += macs[2] * Ho * Wo
    sparse = sparse + macs[2] * Ho * Wo * cs * s3
    if block.downsample is not None:
        dsm = cin * cout * Ho * Wo
        dense += dsm
        sparse = sparse + dsm
    flops = flops + sparse
    perc = sparse / dense

    def push(lst, v):
        v = v.reshape(1)
        return v if lst is None else torch.cat((lst, v))
    return out, push(s3l, s3), push(s2l, s2), push(s1l, s1), push(csl, cs), push(percl, perc), flops


# ------------------------------------------------------------------------------------------------------------------ models
def prepare_for_training(model):
    """Train mode with the BatchNorm STATISTICS frozen (mmdet's norm_eval=True, lad_mmdet_resnet.py:753-758): every BatchNorm runs on its
    running statistics, its affine parameters keep their requires_grad."""
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    return model


def _stem(model, x):
    return model.maxpool(model.relu(model.bn1(model.conv1(x))))          # static stem, laud_resnet.py:316-326 (BatchNorm on frozen statistics)


def train_forward(model, x, temperature=1.0):
    """The reference's forward in training mode on the row kernels.  LAUD-ResNet (`laudnet_amd.ResNet`): -> the 7-tuple (logits, spatial
    sparsity lists of conv3 / conv2 / conv1 per stage, channel sparsity per stage, per-block FLOPs ratios, FLOPs), `laud_resnet.py:312-363`;
    `LAD_MMDet_ResNet`: -> (outs, additional, model_configs), `lad_mmdet_resnet.py:680-751`.  Feed the result to
    `laudnet_amd.sparsity_loss` / the detector's loss as the reference does (`train/main.py:527-604`, `single_stage.py:44-90`)."""
    if not x.is_cuda:
        raise LdnError("laudnet_amd ops need tensors on a HIP device (cuda:N); there is no CPU path")
    is_det = not hasattr(model, "fc")
    h = _stem(model, x)
    # static terms of the FLOPs count (stem conv + its map; the classifier): the module's own shape-only table
    terms, static = model.flops_table(tuple(x.shape))
    head = float(model._head_flops(model.layer4[-1].conv3.out_channels))
    flops = torch.tensor(float(static) - head, device=x.device)
    percl = None
    stage_stats, outs = [], []
    for s in (1, 2, 3, 4):
        state = (h, None, None, None, None, percl, flops)              # sparsity lists restart per stage, the FLOPs ratios and FLOPs run on (:329-347)
        for blk in getattr(model, f"layer{s}"):
            state = block_train(blk, state, temperature)
        h, s3l, s2l, s1l, csl, percl, flops = state
        stage_stats.append((s3l, s2l, s1l, csl))
        outs.append(h)
    s3, s2, s1, cs = ([st[i] for st in stage_stats] for i in range(4))
    if is_det:
        dense = torch.tensor(float(sum(sum(t) for t in terms)) + float(static), device=x.device)     # shape-only (lad_mmdet_resnet.py:691-695)
        additional = {"spatial_sparsity_conv3": s3, "spatial_sparsity_conv2": s2, "spatial_sparsity_conv1": s1, "channel_sparsity": cs,
                      "flops_perc_list": percl, "flops": flops, "dense_flops": dense}
        return tuple(outs), additional, {"dyn_mode": model.dyn_mode, "sparsity_target": model.sparsity_target}
    y = model.fc(torch.flatten(model.avgpool(h), 1))
    flops = flops + head
    return y, s3, s2, s1, cs, percl, flops
