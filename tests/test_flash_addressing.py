"""The LDS image of the fused bilinear kernel (csrc/mi_bilinear_flash.h), restated in numpy: the swizzled image the
LDS-DMA pieces write, the row reads of the score product, the transposed reads (ds_read_b64_tr_b16 semantics, guide T10)
of the output product and the fragment-major layout of the stationary operand.  Every formula below mirrors one in the
header (named in the comments); the test checks that, composed, they deliver exactly the MFMA operand elements DESIGN.md
section 4.1 claims, for the three supported widths.  CPU only: this pins the address ALGEBRA, the compiled kernel is
pinned by the GPU parity tests."""
import numpy as np
import pytest


def swz(row):  # fl_swz
    return ((row & 3) << 2) | ((row >> 2) & 3)


class Cfg:  # FlashCfg<D>
    def __init__(self, d):
        self.D = d
        self.NK, self.NT = d // 16, d // 32
        self.RB = 2 * d                      # bytes per streamed row
        self.CPR = self.RB // 16             # 16-byte chunks per row
        self.RPP = 1024 // self.RB if self.RB < 1024 else 1   # rows per 1 KB piece
        self.PIECES = 32 // self.RPP // 4    # pieces per wave and tile


def piece_u(c, q):  # the wave-uniform swizzle term of piece q (issue_piece / piece_u)
    return swz(q) if c.RPP == 1 else ((8 * (q & 1)) | ((q >> 1) & 3) if c.RPP == 2 else (q & 3))


def lds_image(c, tile):
    """tile: [32, D] element ids.  Returns the LDS stage as an array of element ids (2-byte units)."""
    lds = np.full(32 * c.D, -1, dtype=np.int64)
    src = tile.reshape(-1)
    for q in range(4 * c.PIECES):
        for lane in range(64):
            lr = lane // c.CPR
            vlane = lr * c.RB + 16 * ((lane % c.CPR) ^ ((lr << 2) if c.RPP > 1 else 0))
            s_off = q * 1024 + (vlane ^ (16 * piece_u(c, q)))      # global byte offset inside the tile
            d_off = q * 1024 + 16 * lane                            # the DMA writes a piece linearly
            lds[d_off // 2:d_off // 2 + 8] = src[s_off // 2:s_off // 2 + 8]
    assert (lds >= 0).all()
    return lds


@pytest.mark.parametrize("d", [128, 256, 512])
def test_image_is_the_xor_swizzle_of_guide_t10(d):
    c = Cfg(d)
    tile = np.arange(32 * d).reshape(32, d)
    lds = lds_image(c, tile)
    for row in range(32):
        for cp in range(c.CPR):  # chunk position cp of a row holds source chunk cp ^ swz(row)
            got = lds[(row * c.RB + 16 * cp) // 2:(row * c.RB + 16 * cp) // 2 + 8]
            np.testing.assert_array_equal(got, tile[row, 8 * (cp ^ swz(row)):8 * (cp ^ swz(row)) + 8])


@pytest.mark.parametrize("d", [128, 256, 512])
def test_row_reads_give_the_score_product_operand(d):
    """a0_lane / abase / fl_ring_read_s: MFMA n of the score product wants, on lane (r32, half), the streamed row r32's
    K elements 16 n + 8 half .. + 7."""
    c = Cfg(d)
    tile = np.arange(32 * d).reshape(32, d)
    lds = lds_image(c, tile)
    for lane in range(64):
        r32, half = lane & 31, lane >> 5
        a0_lane = r32 * c.RB + 16 * (half ^ swz(r32))
        for n in range(c.NK):
            addr = (a0_lane ^ (32 * (n & 7))) + 256 * (n >> 3)
            np.testing.assert_array_equal(lds[addr // 2:addr // 2 + 8], tile[r32, 16 * n + 8 * half:16 * n + 8 * half + 8])


def tr_read(lds, addr_of_lane):
    """ds_read_b64_tr_b16: per 16-lane group, lane 4 q + p supplies the address of row q, columns 4 p .. 4 p + 3 of a 4 x 16
    block; lane i of the group receives column i of the four rows (element q = row q)."""
    out = np.zeros((64, 4), dtype=np.int64)
    for g in range(4):
        block = np.zeros((4, 16), dtype=np.int64)
        for q in range(4):
            for p in range(4):
                a = addr_of_lane[16 * g + 4 * q + p]
                assert a % 8 == 0
                block[q, 4 * p:4 * p + 4] = lds[a // 2:a // 2 + 4]
        for i in range(16):
            out[16 * g + i] = block[:, i]
    return out


@pytest.mark.parametrize("d", [128, 256, 512])
def test_transposed_reads_give_the_output_product_operand(d):
    """a1_lane / tbase / fl_ring_read_v: MFMA u = ks * NT + ct of the output product wants, on lane (c = lane & 31, half),
    column 32 ct + c of the streamed rows 16 ks + 4 half + {0..3} (first read) and 16 ks + 8 + 4 half + {0..3} (second):
    the K order of the accumulator layout, which is what makes P usable as the A operand without a shuffle."""
    c = Cfg(d)
    tile = np.arange(32 * d).reshape(32, d)
    lds = lds_image(c, tile)
    lanes = np.arange(64)
    half, g16, q4, p4 = lanes >> 5, lanes >> 4, (lanes & 15) >> 2, lanes & 3
    cl = 2 * (g16 & 1) + (p4 >> 1)
    a1_lane = (4 * half + q4) * c.RB + 16 * ((cl ^ half) | (q4 << 2)) + 8 * (p4 & 1)
    for ks in range(2):
        for ct in range(c.NT):
            for j in range(2):
                v = 2 * (ct & 3) + j
                tbase = a1_lane ^ (16 * ((4 * (v >> 1)) ^ (2 * (v & 1))))
                addr = tbase + 256 * (ct >> 2) + (16 * ks + 8 * j) * c.RB
                got = tr_read(lds, addr)
                for lane in range(64):
                    rows = 16 * ks + 8 * j + 4 * (lane >> 5) + np.arange(4)
                    np.testing.assert_array_equal(got[lane], tile[rows, 32 * ct + (lane & 31)])


@pytest.mark.parametrize("d", [128, 256, 512])
def test_fragment_major_layout_matches_the_stationary_loads(d):
    """frag_major_offset (mi_gemm_bf16.h) against the kernel's Q loads: qf[kk] = 8 elements at
    ((row block * NK + kk) * 64 + lane) * 8, lane = r32 + 32 half, holding Q[row][16 kk + 8 half .. + 7]."""
    c = Cfg(d)
    m = 96
    q = np.arange(m * d).reshape(m, d)
    frag = np.zeros(m * d, dtype=np.int64)
    for row in range(m):
        for col8 in range(0, d, 8):
            off = (((row >> 5) * (d >> 4) + (col8 >> 4)) * 64 + ((col8 >> 3) & 1) * 32 + (row & 31)) * 8
            frag[off:off + 8] = q[row, col8:col8 + 8]
    for gi in range(m):
        for half in range(2):
            lane = (gi & 31) + 32 * half
            for kk in range(c.NK):
                off = ((gi >> 5) * c.NK * 64 + lane) * 8 + kk * 512
                np.testing.assert_array_equal(frag[off:off + 8], q[gi, 16 * kk + 8 * half:16 * kk + 8 * half + 8])
