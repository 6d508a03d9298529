"""Sparse storage of a fixed (untrained) coupling matrix with DENSE-IDENTICAL arithmetic.

The reference multiplies densely even when `utility.random_connectivity(p=0.1)` built the matrix
(rectipy/utility.py:153-178), so a step streams 4 n^2 bytes of mostly zeros.  The dense step kernel gives every
(lane l, component c) of a wave the columns 256 m + 4 l + c of a row and accumulates them in a fixed chunk order
(DESIGN.md "Summation order"); a zero entry leaves that accumulator untouched (fma(0, r, a) == a).  The packed layout
therefore keeps, per row and per LANE, only the nonzeros of the lane's four components, in the order the dense kernel
meets them (chunk order of that row; inside a chunk c = 0..3), four entries per lane per group:

    vals[(rowoff[i] + k) * 256 + 4 l + e]   entry 4 k + e of lane l of row i (0.0 past the end of the lane's list)
    code[(rowoff[i] + k) * 256 + 4 l + e]   (c << 14) | m : its component and column chunk, column = 256 m + 4 l + c
    rowoff[i + 1] - rowoff[i] = L_i          ceil(longest lane list of row i / 4)

Each of the 256 accumulators of a row still receives exactly its own nonzeros in the dense order, so the sparse step
kernel reproduces the dense result bit for bit while streaming 6 bytes per stored entry.  Only the 16-byte path
(n % 4 == 0) has a packed form; chunk indices take 14 bits (n < 4.2 million).
"""
from typing import NamedTuple, Optional

import torch

STAGGER = 3  # kStagger of rp_dot.h: start chunk m0 = (3 * (global_row >> 3)) mod M
WINDOW_CHUNKS = 128     # 128 chunks = 32768 columns = the 128 KiB of the source vector one workgroup keeps in LDS
WINDOW_COLUMNS = WINDOW_CHUNKS * 256
BATCH_ROWS = 64         # rows per workgroup pass of the windowed kernel (16 waves x 4 rows)


BLOCK_ROWS = 8          # default rows per block of the blocked form (pack_rows_blocked); the kernel's LDS stash holds
                        # block_rows x 64 floats per wave (16 waves x 8 rows = 32 KiB next to the 128 KiB source copy at n = 32768)
END_FLAG = 0x2000       # blocked form: bit 13 of a code marks the last entry of the lane's segment of a row


class PackedRows(NamedTuple):
    vals: torch.Tensor    # float32 [groups * 256]
    code: torch.Tensor    # int16 (bit pattern of uint16) [groups * 256]
    rowoff: torch.Tensor  # int32 [rows + 1], in groups of 256 entries  (windowed: [slots * phases + 1])
    density: float
    stored_bytes: int
    # windowed form (n > WINDOW_COLUMNS): rows regrouped by the window their sweep starts in, entries split by phase
    row_ids: Optional[torch.Tensor] = None       # int32 [batches * BATCH_ROWS]: row of each slot, -1 = padding
    batch_class: Optional[torch.Tensor] = None   # int32 [batches]: start window of the rows of a batch
    phase_window: Optional[torch.Tensor] = None  # int32 [windows * phases]: window staged in phase ph of class c, -1 = none
    phases: int = 0
    batches: int = 0
    block_rows: int = 0   # > 0: blocked form (pack_rows_blocked), rowoff = [blocks + 1]


def density(W: torch.Tensor) -> float:
    return float((W != 0).float().mean().item()) if W.numel() else 1.0


def pack_rows(W: torch.Tensor, row0: int = 0, block_rows: int = 512) -> PackedRows:
    """Pack the rows of W ([rows, n], n % 4 == 0) in dense traversal order; `row0` = global index of row 0."""
    rows, n = W.shape
    if n % 4:
        raise ValueError("the packed layout exists for n % 4 == 0 only")
    dev = W.device
    M = n >> 8                      # full 256-column chunks
    Mp = M + (1 if n & 255 else 0)  # + the partial chunk, visited last
    if Mp > (1 << 14):
        raise ValueError("n too large for 14-bit chunk indices")
    lens, chunks = [], []
    nnz = 0
    q = torch.arange(Mp, device=dev)
    comp = torch.arange(4, device=dev, dtype=torch.int32)
    for r0 in range(0, rows, block_rows):
        r1 = min(r0 + block_rows, rows)
        B = r1 - r0
        Wb = W[r0:r1].to(torch.float32)
        if Mp * 256 != n:
            Wb = torch.nn.functional.pad(Wb, (0, Mp * 256 - n))
        Wb = Wb.reshape(B, Mp, 256)
        # chunk visited at position q of row i: (m0_i + q) mod M for q < M, then the partial chunk M
        g = (torch.arange(r0, r1, device=dev) + row0) >> 3
        m0 = (STAGGER * g) % M if M > 0 else torch.zeros_like(g)
        order = torch.where(q[None, :] < M, (m0[:, None] + q[None, :]) % max(M, 1), q[None, :])   # [B, Mp]
        Wrot = torch.gather(Wb, 1, order[:, :, None].expand(B, Mp, 256))
        # a lane's sequence: position q major, component c minor -> [B, Mp * 4, 64]
        seq = Wrot.reshape(B, Mp, 64, 4).permute(0, 1, 3, 2).reshape(B, Mp * 4, 64)
        nz = seq != 0
        rank = nz.cumsum(dim=1, dtype=torch.int32) - 1                                               # [B, Mp*4, 64]
        L = ((rank[:, -1, :] + 1).max(dim=1).values + 3) // 4                                        # groups per row
        off = torch.cumsum(L, 0) - L                                                                 # block-local
        groups = int(L.sum().item())
        vals = torch.zeros(groups * 256, dtype=torch.float32, device=dev)
        code = torch.zeros(groups * 256, dtype=torch.int16, device=dev)
        bi, ji, li = nz.nonzero(as_tuple=True)
        r = rank[bi, ji, li].to(torch.int64)
        flat = (off[bi].to(torch.int64) + r // 4) * 256 + 4 * li + r % 4
        vals[flat] = seq[bi, ji, li]
        word = (comp[ji % 4] << 14) | order[bi, ji // 4].to(torch.int32)
        code[flat] = word.to(torch.int16)  # keeps the low 16 bits
        nnz += int(bi.numel())
        lens.append(L.to(torch.int32))
        chunks.append((vals, code))
    L_all = torch.cat(lens)
    rowoff = torch.zeros(rows + 1, dtype=torch.int32, device=dev)
    rowoff[1:] = torch.cumsum(L_all, 0).to(torch.int32)
    vals = torch.cat([c[0] for c in chunks]) if chunks else torch.zeros(0, device=dev)
    code = torch.cat([c[1] for c in chunks]) if chunks else torch.zeros(0, dtype=torch.int16, device=dev)
    if vals.numel() == 0:  # an all-zero matrix: keep one (unused) group so that the pointers are valid
        vals = torch.zeros(256, dtype=torch.float32, device=dev)
        code = torch.zeros(256, dtype=torch.int16, device=dev)
    return PackedRows(vals.contiguous(), code.contiguous(), rowoff, nnz / max(1, rows * n),
                      int(vals.numel()) * 6 + int(rowoff.numel()) * 4)


def pack_rows_blocked(W: torch.Tensor, row0: int = 0, block_rows: int = BLOCK_ROWS, proc_rows: int = 512) -> PackedRows:
    """Blocked form (n <= WINDOW_COLUMNS): the per-row form pads every row to its longest lane list -- 1.35x the ideal
    bytes at p = 0.1, n = 32768 (64 lists of Binomial(512, p) entries each).  Here `block_rows` consecutive rows share one
    padded stream: lane l's list of a block is the concatenation of its lists of the block's rows, every row segment
    closed by an entry whose code carries END_FLAG (an empty segment is one dummy entry: value 0, code END_FLAG), and only
    the BLOCK is padded to its longest lane list.  Lanes move from row to row at their own pace; at END a lane parks its
    lane sum (a0+a1)+(a2+a3) of that row in an LDS slot and clears its accumulators, the butterflies of the block's rows
    run when the block's stream is exhausted.  Every accumulator still sees exactly its nonzeros in the dense order, the
    lane sums and butterflies are unchanged: bit-identical output; n = 32768, p = 0.1: 1.116x the ideal bytes at
    block_rows = 8, 1.078x at 16 (per-row form: 1.354x).

        vals/code[(rowoff[b] + k) * 256 + 4 l + e]   entry 4 k + e of lane l of block b
        code = (c << 14) | END_FLAG? | m              (chunk m < 8192)
    """
    rows, n = W.shape
    if n % 4:
        raise ValueError("the packed layout exists for n % 4 == 0 only")
    if proc_rows % block_rows:
        raise ValueError("proc_rows must be a multiple of block_rows")
    dev = W.device
    M = n >> 8
    Mp = M + (1 if n & 255 else 0)
    if Mp > END_FLAG:
        raise ValueError("n too large for the blocked form (13-bit chunk indices)")
    q = torch.arange(Mp, device=dev)
    comp = torch.arange(4, device=dev, dtype=torch.int32)
    lens, chunks, nnz = [], [], 0
    for r0 in range(0, rows, proc_rows):
        r1 = min(r0 + proc_rows, rows)
        B = r1 - r0
        Wb = W[r0:r1].to(torch.float32)
        if Mp * 256 != n:
            Wb = torch.nn.functional.pad(Wb, (0, Mp * 256 - n))
        Wb = Wb.reshape(B, Mp, 256)
        g = (torch.arange(r0, r1, device=dev) + row0) >> 3
        m0 = (STAGGER * g) % M if M > 0 else torch.zeros_like(g)
        order = torch.where(q[None, :] < M, (m0[:, None] + q[None, :]) % max(M, 1), q[None, :])
        Wrot = torch.gather(Wb, 1, order[:, :, None].expand(B, Mp, 256))
        seq = Wrot.reshape(B, Mp, 64, 4).permute(0, 1, 3, 2).reshape(B, Mp * 4, 64)
        nz = seq != 0
        rank = nz.cumsum(dim=1, dtype=torch.int32) - 1                       # [B, Mp*4, 64] rank inside (row, lane)
        cnt = rank[:, -1, :] + 1                                              # [B, 64] nonzeros of (row, lane)
        seg = cnt.clamp(min=1)                                                # an empty segment holds one dummy entry
        nb = -(-B // block_rows)
        pad_rows = nb * block_rows - B
        seg_b = torch.nn.functional.pad(seg, (0, 0, 0, pad_rows)).reshape(nb, block_rows, 64)
        start = (torch.cumsum(seg_b, 1) - seg_b).reshape(nb * block_rows, 64)[:B]   # first position of (row, lane)
        L = (seg_b.sum(1).max(dim=1).values + 3) // 4                         # [nb] groups per block
        off = torch.cumsum(L, 0) - L
        groups = int(L.sum().item())
        vals = torch.zeros(groups * 256, dtype=torch.float32, device=dev)
        code = torch.zeros(groups * 256, dtype=torch.int16, device=dev)
        blk = torch.arange(B, device=dev) // block_rows
        bi, ji, li = nz.nonzero(as_tuple=True)
        r = rank[bi, ji, li]
        pos = (start[bi, li] + r).to(torch.int64)
        flat = (off[blk[bi]].to(torch.int64) + pos // 4) * 256 + 4 * li + pos % 4
        vals[flat] = seq[bi, ji, li]
        word = (comp[ji % 4] << 14) | order[bi, ji // 4].to(torch.int32)
        word = word | torch.where(r == cnt[bi, li] - 1, END_FLAG, 0).to(torch.int32)
        code[flat] = word.to(torch.int16)
        eb, el = (cnt == 0).nonzero(as_tuple=True)                           # dummy entries closing empty segments
        if eb.numel():
            pos = start[eb, el].to(torch.int64)
            flat = (off[blk[eb]].to(torch.int64) + pos // 4) * 256 + 4 * el + pos % 4
            code[flat] = torch.tensor(END_FLAG, dtype=torch.int32).to(torch.int16)
        nnz += int(bi.numel())
        lens.append(L.to(torch.int32))
        chunks.append((vals, code))
    L_all = torch.cat(lens) if lens else torch.zeros(0, dtype=torch.int32, device=dev)
    rowoff = torch.zeros(int(L_all.numel()) + 1, dtype=torch.int32, device=dev)
    rowoff[1:] = torch.cumsum(L_all, 0).to(torch.int32)
    vals = torch.cat([c[0] for c in chunks]) if chunks else torch.zeros(0, device=dev)
    code = torch.cat([c[1] for c in chunks]) if chunks else torch.zeros(0, dtype=torch.int16, device=dev)
    if vals.numel() == 0:
        vals = torch.zeros(256, dtype=torch.float32, device=dev)
        code = torch.zeros(256, dtype=torch.int16, device=dev)
    return PackedRows(vals.contiguous(), code.contiguous(), rowoff, nnz / max(1, rows * n),
                      int(vals.numel()) * 6 + int(rowoff.numel()) * 4, block_rows=block_rows)


def pack_rows_windowed(W: torch.Tensor, row0: int = 0, block_rows: int = 256) -> PackedRows:
    """Packed form for n > WINDOW_COLUMNS.  The source vector no longer fits in LDS, so the sweep of a row is cut into
    PHASES, each confined to one window of WINDOW_CHUNKS chunks: a row that starts in window w0 visits
    w0 (its tail), w0+1, ..., last, 0, ..., w0 (its head) and finally the window of the partial chunk.  Rows are
    regrouped into batches of BATCH_ROWS rows with the same w0, so that a workgroup stages one window at a time for a
    whole batch; a lane's entries are stored per phase (rowoff[slot * phases + ph]), still in dense traversal order, and
    a code holds the chunk index RELATIVE to its window.  Same arithmetic, same bits."""
    rows, n = W.shape
    if n % 4:
        raise ValueError("the packed layout exists for n % 4 == 0 only")
    dev = W.device
    Wc = WINDOW_CHUNKS
    M = n >> 8
    Mp = M + (1 if n & 255 else 0)
    Pw = max(1, -(-M // Wc))          # windows that hold full chunks
    wp = M // Wc                       # window of the partial chunk (== Pw when M is a multiple of Wc)
    NP = Pw + 2
    g = (torch.arange(rows, device=dev) + row0) >> 3
    m0_all = (STAGGER * g) % M if M > 0 else torch.zeros_like(g)
    cls_all = m0_all // Wc
    ids = []
    classes = []
    for c in range(Pw):
        sel = torch.nonzero(cls_all == c, as_tuple=True)[0]
        pad = (-int(sel.numel())) % BATCH_ROWS
        ids.append(sel.to(torch.int32))
        ids.append(torch.full((pad,), -1, dtype=torch.int32, device=dev))
        classes.append(torch.full(((int(sel.numel()) + pad) // BATCH_ROWS,), c, dtype=torch.int32, device=dev))
    row_ids = torch.cat(ids)
    batch_class = torch.cat(classes)
    slots = int(row_ids.numel())
    sched = torch.full((Pw, NP), -1, dtype=torch.int32)
    for c in range(Pw):
        for d in range(Pw):
            sched[c, d] = (c + d) % Pw
        sched[c, Pw] = c
        if Mp > M:
            sched[c, Pw + 1] = wp      # (when wp == c the partial chunk shares phase Pw; the extra phase stays empty)
    q = torch.arange(Mp, device=dev)
    comp = torch.arange(4, device=dev, dtype=torch.int32)
    lens, chunks, nnz = [], [], 0
    for s0 in range(0, slots, block_rows):
        s1 = min(s0 + block_rows, slots)
        B = s1 - s0
        rid = row_ids[s0:s1].to(torch.int64)
        valid = rid >= 0
        ridc = rid.clamp(min=0)
        Wb = W[ridc].to(torch.float32) * valid[:, None]
        if Mp * 256 != n:
            Wb = torch.nn.functional.pad(Wb, (0, Mp * 256 - n))
        Wb = Wb.reshape(B, Mp, 256)
        m0 = m0_all[ridc]
        c0 = cls_all[ridc]
        order = torch.where(q[None, :] < M, (m0[:, None] + q[None, :]) % max(M, 1), q[None, :])   # chunk at position q
        Wrot = torch.gather(Wb, 1, order[:, :, None].expand(B, Mp, 256))
        w = order // Wc
        ph = torch.where((w == c0[:, None]) & (order < m0[:, None]), torch.full_like(w, Pw), (w - c0[:, None]) % Pw)
        if Mp > M:
            ph[:, M] = torch.where(c0 == wp, torch.full_like(c0, Pw), torch.full_like(c0, Pw + 1))
        seq = Wrot.reshape(B, Mp, 64, 4).permute(0, 1, 3, 2).reshape(B, Mp * 4, 64)
        phj = ph.repeat_interleave(4, dim=1)                                                        # [B, Mp*4]
        nz = seq != 0
        rank = nz.cumsum(dim=1, dtype=torch.int32) - 1                                               # over the whole row
        cnt = torch.zeros(B, NP, 64, dtype=torch.int32, device=dev)
        cnt.scatter_add_(1, phj[:, :, None].expand(B, Mp * 4, 64), nz.to(torch.int32))
        start = torch.cumsum(cnt, 1) - cnt                                                           # entries before a phase
        L = (cnt.max(dim=2).values + 3) // 4                                                         # [B, NP] groups
        Lf = L.reshape(-1)
        off = (torch.cumsum(Lf, 0) - Lf).reshape(B, NP)
        groups = int(Lf.sum().item())
        vals = torch.zeros(groups * 256, dtype=torch.float32, device=dev)
        code = torch.zeros(groups * 256, dtype=torch.int16, device=dev)
        bi, ji, li = nz.nonzero(as_tuple=True)
        pj = phj[bi, ji]
        r = (rank[bi, ji, li] - start[bi, pj, li]).to(torch.int64)
        flat = (off[bi, pj].to(torch.int64) + r // 4) * 256 + 4 * li + r % 4
        vals[flat] = seq[bi, ji, li]
        word = (comp[ji % 4] << 14) | (order[bi, ji // 4].to(torch.int32) & (Wc - 1))
        code[flat] = word.to(torch.int16)
        nnz += int(bi.numel())
        lens.append(Lf.to(torch.int32))
        chunks.append((vals, code))
    L_all = torch.cat(lens)
    rowoff = torch.zeros(slots * NP + 1, dtype=torch.int32, device=dev)
    rowoff[1:] = torch.cumsum(L_all, 0).to(torch.int32)
    vals = torch.cat([c[0] for c in chunks])
    code = torch.cat([c[1] for c in chunks])
    if vals.numel() == 0:
        vals = torch.zeros(256, dtype=torch.float32, device=dev)
        code = torch.zeros(256, dtype=torch.int16, device=dev)
    stored = int(vals.numel()) * 6 + int(rowoff.numel()) * 4 + slots * 4
    return PackedRows(vals.contiguous(), code.contiguous(), rowoff, nnz / max(1, rows * n), stored,
                      row_ids=row_ids.contiguous(), batch_class=batch_class.contiguous(),
                      phase_window=sched.reshape(-1).to(dev).contiguous(), phases=NP, batches=int(batch_class.numel()))
