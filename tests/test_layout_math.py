"""Index math of the HIP kernels, restated in Python and checked exhaustively on the CPU: LDS swizzles are bijective
and conflict-free for the access patterns the kernels use, the XCD-aware block order is a permutation, the GEGLU /
transposing-read lane maps are what the kernels assume.  (The kernels themselves only run on the MI355X; these tests
pin the layout invariants their comments claim, so an edit that breaks one fails here first.)"""
import itertools

import pytest


def banks_of(addr_bytes, nbytes, modulus):
    return [((addr_bytes // 4) + i) % modulus for i in range(nbytes // 4)]


def worst_conflict(accesses, modulus):
    """accesses: list of (byte address, byte width) issued in one LDS cycle group -> max distinct addresses per bank."""
    per_bank = {}
    for a, n in accesses:
        for b, dw in zip(banks_of(a, n, modulus), range(n // 4)):
            per_bank.setdefault(b, set()).add(a + 4 * dw)
    return max(len(v) for v in per_bank.values())


# ---------------------------------------------------------------- conv3x3.hip: LDS-DMA source swizzle vs fragment reads
@pytest.mark.parametrize("BK,rows", [(64, 256), (64, 320), (32, 128)])
def test_conv_operand_tile_swizzle(BK, rows):
    """global_load_lds writes lane-linear: chunk position c of tile row R receives DATA chunk c ^ swz(R)
    (conv3x3.hip: pd8 / bptr); the MFMA fragment read of (row R, data chunk d) goes to position d ^ swz(R)
    (koff).  Check they meet, and that the 16 rows x 16 bytes one ds_read_b128 lane group touches are conflict-free."""
    CPR, ROWB = BK // 8, BK * 2
    RPB = 256 // ROWB
    swz = lambda r: (r // RPB) & (CPR - 1)
    lds = {}
    for r in range(rows):
        for pos in range(CPR):
            lds[r * ROWB + pos * 16] = (r, pos ^ swz(r))               # what the DMA put there
    for r in range(rows):
        for d in range(CPR):
            assert lds[r * ROWB + ((d ^ swz(r)) * 16)] == (r, d)
    # ds_read_b128 is serviced in 4 groups of 16 lanes (MI355X_MICROARCH.md, LDS); lanes of a group read 16
    # consecutive rows at one data chunk: 16 B x 16 = 256 B = every one of the 64 banks exactly once
    for base in range(0, rows - 15, 16):
        for d in range(CPR):
            acc = [(r * ROWB + ((d ^ swz(r)) * 16), 16) for r in range(base, base + 16)]
            assert worst_conflict(acc, 64) == 1


def test_conv_fragment_offsets_are_base_plus_immediate():
    """koff[kc] is shared by every 32-row block of a wave (swz only looks at row bits below 32)."""
    for BK in (32, 64):
        CPR, RPB = BK // 8, 256 // (BK * 2)
        for col, blk, wave_row0 in itertools.product(range(32), range(5), (0, 160, 64, 128)):
            r = wave_row0 + blk * 32 + col
            assert (r // RPB) & (CPR - 1) == (col // RPB) & (CPR - 1)


# ---------------------------------------------------------------- conv3x3.hip epilogue: swizzled 16-byte pieces
@pytest.mark.parametrize("ROWB", [320, 128])
def test_conv_epilogue_transpose_swizzle(ROWB):
    pieces = ROWB // 16
    px = (lambda r: r & 7) if ROWB == 128 else (lambda r: (r >> 1) & 3)
    slot = lambda row, p16, half: row * ROWB + ((p16 ^ px(row)) << 4) + ((half ^ ((row >> 3) & 1)) << 3)
    # bijective over a 32-pixel block
    seen = {slot(r, p, h) for r in range(32) for p in range(pieces) for h in range(2)}
    assert len(seen) == 32 * pieces * 2 and max(seen) < 32 * ROWB
    # fragment side: ds_write_b64, 16 consecutive pixels (lanes) x one (piece, half): 2-dword writes, 32-bank modulus
    for base in (0, 16):
        for p, h in itertools.product(range(pieces), range(2)):
            assert worst_conflict([(slot(r, p, h), 8) for r in range(base, base + 16)], 32) == 1
    # row-major side reads whole pieces: piece p of row r sits at (p ^ px(r)) with its halves swapped on odd octets
    for r, p in itertools.product(range(32), range(pieces)):
        lo, hi = slot(r, p, 0), slot(r, p, 1)
        assert {lo, hi} == {r * ROWB + ((p ^ px(r)) << 4), r * ROWB + ((p ^ px(r)) << 4) + 8}
        assert (lo > hi) == bool((r >> 3) & 1)


# ---------------------------------------------------------------- attn_fwd.hip: XCD-aware block order
@pytest.mark.parametrize("nb", [1, 7, 8, 9, 63, 64, 100, 5120, 12801])
def test_xcd_block_remap_is_a_permutation(nb):
    """blockIdx -> logical tile (attn_fwd.hip / conv3x3.hip): XCD x (= blockIdx % 8) owns one contiguous range."""
    qn, rn = nb // 8, nb % 8
    out = []
    for lb in range(nb):
        xcd, idx = lb % 8, lb // 8
        out.append((xcd * (qn + 1) if xcd < rn else rn * (qn + 1) + (xcd - rn) * qn) + idx)
    assert sorted(out) == list(range(nb))
    for xcd in range(8):
        mine = sorted(o for lb, o in enumerate(out) if lb % 8 == xcd)
        assert mine == list(range(mine[0], mine[0] + len(mine))) if mine else True


# ---------------------------------------------------------------- attn_fwd.hip / temporal_attn.hip: transposing LDS read
def tr16_gather(lane_chunk_addr):
    """ds_read_b64_tr_b16 as probed by tools/tr_probe.hip: within a 16-lane group, lane i receives, for j = 0..3,
    element (i & 3) of the 8-byte chunk addressed by lane 4 j + (i >> 2)."""
    res = {}
    for lane in range(64):
        g, i = lane // 16, lane % 16
        res[lane] = [lane_chunk_addr[g * 16 + 4 * j + (i >> 2)] + 2 * (i & 3) for j in range(4)]
    return res


@pytest.mark.parametrize("D,VP", [(64, 96), (32, 32)])
def test_attention_v_gather_matches_pv_operand(D, VP):
    """The PV MFMA (32x32x16, A = V^T) of key chunk c wants lane (channel = l & 31, hi = l >> 5) to hold keys
    16 c + 4 hi + {0..3} and 16 c + 8 + 4 hi + {0..3} of that channel -- the C-fragment key order of S^T.  Two
    transposing reads from the row-major V tile (pitch VP elements) deliver exactly that, conflict-free."""
    for kb, c, dvb in itertools.product(range(2), range(2), range(D // 32)):
        for u in range(2):
            addr = {}
            for lane in range(64):
                hi, half, l16 = lane >> 5, (lane >> 4) & 1, lane & 15
                vfrag = (4 * hi + (l16 >> 2)) * VP + 16 * half + 4 * (l16 & 3)
                addr[lane] = 2 * (vfrag + (kb * 32 + 16 * c + 8 * u) * VP + dvb * 32)
            got = tr16_gather(addr)
            for lane in range(64):
                hi, ch = lane >> 5, dvb * 32 + (lane & 31)
                want = [2 * ((kb * 32 + 16 * c + 8 * u + 4 * hi + j) * VP + ch) for j in range(4)]
                assert got[lane] == want
            # bank behaviour of the 32-lane halves (64-bank modulus for tr reads): each 8-byte chunk is read once
            for half_lanes in (range(0, 32), range(32, 64)):
                assert worst_conflict([(addr[l], 8) for l in half_lanes], 64) == 1


def test_temporal_v_gather_matches_pv_operand():
    """temporal_attn_mfma_kernel: 16x16x16 MFMA, A = V^T: lane (channel = l & 15, g = l >> 4) needs keys 4g .. 4g+3."""
    for G, d, hl, c0 in [(320, 40, 3, 16), (320, 160, 1, 144), (64, 8, 5, 0)]:
        pitch = 3 * G + 8
        addr = {lane: 2 * ((4 * (lane >> 4) + ((lane & 15) >> 2)) * pitch + 2 * G + hl * d + c0 + 4 * (lane & 3))
                for lane in range(64)}
        got = tr16_gather(addr)
        for lane in range(64):
            g, ch = lane >> 4, c0 + (lane & 15)
            assert got[lane] == [2 * ((4 * g + j) * pitch + 2 * G + hl * d + ch) for j in range(4)]


def test_temporal_row_pitch_spreads_frames_over_banks():
    """16 frame rows, 16-byte fragment reads at the same channel offset: pitch 3G + 8 elements puts them 4 banks apart."""
    G = 320
    pitch_b = (3 * G + 8) * 2
    for off in range(0, 3 * G * 2, 16):
        assert worst_conflict([(f * pitch_b + off, 16) for f in range(16)], 64) == 1


# ---------------------------------------------------------------- MFMA C-fragment key order used by the softmax
def test_mfma32_c_fragment_rows():
    rows = lambda r, hi: (r & 3) + 8 * (r >> 2) + 4 * hi
    for hi in range(2):
        assert sorted(rows(r, hi) for r in range(16)) == sorted(set(range(32)) - {rows(r, 1 - hi) for r in range(16)})
    # registers 8c .. 8c+7 of lane-half hi are keys 16c + {0..3} + 4 hi and 16c + 8 + {0..3} + 4 hi (PV operand order)
    for c, hi in itertools.product(range(2), range(2)):
        assert [rows(r, hi) for r in range(8 * c, 8 * c + 8)] == \
            [16 * c + 4 * hi + j for j in range(4)] + [16 * c + 8 + 4 * hi + j for j in range(4)]


# ---------------------------------------------------------------- attn_fwd.hip: resident-K/V cross attention (xattn_resident_kernel)
@pytest.mark.parametrize("Bkv,H,kv_group,Nq,nwv", [(40, 5, 16, 1024, 12), (2, 5, 16, 8192, 12), (3, 4, 2, 320, 4), (1, 5, 1, 1024, 4),
                                                  (2, 20, 16, 32, 4), (1, 1, 1, 32, 4), (3, 5, 2, 1056, 4), (4, 5, 16, 1024, 12)])
def test_xattn_block_walk_covers_every_query_block_once(Bkv, H, kv_group, Nq, nwv):
    """The kernel deals contiguous ranges of the (pair, frame, 32-row block) sequence to workgroups, segments a range at pair
    boundaries (K / V re-staged), strides a segment over the workgroup's waves and advances (frame, block) without divisions;
    the LDS-DMA variant requests two blocks ahead and re-requests the wave's last block past the end.  Every block is processed
    exactly once, by the slot its request went to, and nothing ever indexes past a video's frames."""
    nqb = (Nq + 31) // 32
    bpp = kv_group * nqb
    total = Bkv * H * bpp
    per = min(max(total // 1024, 8), 48)
    G = 256 * -(-total // (144 * 256)) if nwv == 12 else -(-total // per)
    seen = set()
    for w in range(G):
        r0, r1 = total * w // G, total * (w + 1) // G
        j = r0
        while j < r1:
            pair = j // bpp
            seg_end, base = min(r1, (pair + 1) * bpp), pair * bpp
            for wid in range(nwv):
                jj = j + wid
                g, qb = divmod(jj - base, nqb)
                # request stream of the ring variant: (block index requested, slot)
                req, jr, gr, qr = [], jj, g, qb
                def request(slot):
                    nonlocal jr, gr, qr
                    req.append(((gr, qr), slot))
                    if jr + nwv < seg_end:
                        jr += nwv
                        qr += nwv
                        while qr >= nqb:
                            qr -= nqb
                            gr += 1
                if jj < seg_end:
                    request(0)
                    request(1)
                it = 0
                while jj < seg_end:
                    assert 0 <= g < kv_group and 0 <= qb < nqb and (g, qb) == divmod(jj - base, nqb)
                    # the newest request into slot it & 1 before this iteration's refill is this block's
                    assert [r for r, s in req if s == (it & 1)][-1] == (g, qb)
                    assert (pair, g, qb) not in seen
                    seen.add((pair, g, qb))
                    request(it & 1)
                    if jj + nwv < seg_end:
                        qb += nwv
                        while qb >= nqb:
                            qb -= nqb
                            g += 1
                    jj += nwv
                    it += 1
            j = seg_end
    assert len(seen) == total


def test_xattn_query_ring_swizzle_is_conflict_free():
    """global_load_lds piece i, lane l lands at ring byte (64 i + l) * 16 = row (8 i + l / 8), chunk position l % 8, and was
    given data chunk (l % 8) ^ ((row >> 1) & 7) of that query row as its source; the fragment read of lane (col, hi), k-step dc
    goes to position (2 dc + hi) ^ ((col >> 1) & 7) of row col.  They meet, every row arrives as one whole 128-byte line, and
    each 16-lane group of the ds_read_b128 touches all 64 banks once."""
    lds = {}
    for i in range(4):
        for lane in range(64):
            row, pos = 8 * i + lane // 8, lane % 8
            assert (64 * i + lane) * 16 == row * 128 + pos * 16
            lds[(row, pos)] = pos ^ ((row >> 1) & 7)                     # data chunk stored there
        for row in range(8 * i, 8 * i + 8):
            assert sorted(lds[(row, p)] for p in range(8)) == list(range(8))
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for dc in range(4):
        for lane in range(64):
            col, hi = lane & 31, lane >> 5
            assert lds[(col, (2 * dc + hi) ^ ((col >> 1) & 7))] == 2 * dc + hi
        for grp in groups:
            acc = [((l & 31) * 128 + (((2 * dc + (l >> 5)) ^ (((l & 31) >> 1) & 7)) << 4), 16) for l in grp]
            assert worst_conflict(acc, 64) == 1


def test_xattn_wide_store_exchange():
    """v_permlane32_swap(first, second) exchanges lanes 32-63 of `first` with lanes 0-31 of `second`.  With first = this lane's
    8 bytes of column group g, second = its 8 bytes of group g + 1 (lane (col, hi) holds channels 8 g' + 4 hi .. + 3), every
    lane ends up with the 16 contiguous bytes of channels 8 (g + hi) .. + 7 of its query row, low half first."""
    for g in (0, 2):
        first = {l: ("g", g, l >> 5, l & 31) for l in range(64)}           # (group, hi of the owner, query col)
        second = {l: ("g", g + 1, l >> 5, l & 31) for l in range(64)}
        f2, s2 = dict(first), dict(second)
        for l in range(32):
            f2[l + 32], s2[l] = second[l], first[l + 32]
        for l in range(64):
            col, hi = l & 31, l >> 5
            lo, hi_half = f2[l], s2[l]
            grp = g + hi
            assert lo == ("g", grp, 0, col) and hi_half == ("g", grp, 1, col)    # channels 8 grp + 0..3 then + 4..7, same row
