"""TEST INFRASTRUCTURE: numpy stand-in for orip_stream_codes (the HIP kernel) so that the host logic of orip/stream.py can be checked on the CPU
against the reference's golden bytes.  Same closed form as csrc/stream.hip; itself pinned by the `bres*` golden vectors."""
import numpy as np


def codes_numpy(moves):
    m = np.asarray(moves, np.int64).reshape(-1, 4)
    dx, dy = np.abs(m[:, 2] - m[:, 0]), np.abs(m[:, 3] - m[:, 1])
    cnt = np.maximum(dx, dy)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
    seg = np.repeat(np.arange(len(m)), cnt)
    k = np.arange(int(off[-1])) - off[seg]
    DX, DY = dx[seg], dy[seg]
    xpos, ypos = (m[:, 0] < m[:, 2])[seg], (m[:, 1] < m[:, 3])[seg]

    def cdiv0(a, b):
        return np.where(a <= 0, 0, -((-a) // np.maximum(b, 1)))
    xmaj = DX >= DY
    mx = np.where(xmaj, True, cdiv0(2 * (k + 1) * DX - DY, 2 * DY) != cdiv0(2 * k * DX - DY, 2 * DY))
    my = np.where(xmaj, cdiv0(2 * (k + 1) * DY - DX, 2 * DX) != cdiv0(2 * k * DY - DX, 2 * DX), True)
    diag = np.where(xpos, np.where(ypos, 1, 3), np.where(ypos, 7, 5))
    c = np.where(mx & my, diag, np.where(mx, np.where(xpos, 2, 6), np.where(ypos, 0, 4)))
    return off, c.astype(np.uint8)
