// Host-only check of the device weight layouts (rlcontrol_amd/csrc/rlc_common.h): rlc_blk_index is a bijection
// onto the padded block grid, pack/unpack round-trip the compact ABI blob under both layouts, Wc2's action rows
// land in their own block row and the trunk padding rows stay zero.  Built and run by tests/test_layout_host.py.
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "rlc_common.h"

void rlc_set_error(const char*, ...) {}

static int check_dims(int S, int A, int H1, int HA, int HC, int B) {
    for (int blocked = 0; blocked <= 1; blocked++) {
        const RlcDims d = rlc_make_dims(S, A, H1, HA, HC, B, blocked);
        if (d.Pdev > d.Ppad || (d.Ppad & 63)) { printf("bad stride\n"); return 1; }
        for (int i = 0; i < 10; i++)
            if (d.seg_dev[i] & 63) { printf("segment %d not 256-byte aligned\n", i); return 1; }
        std::vector<float> compact(d.P), padded(d.Ppad, 0.0f), back(d.P, -1.0f);
        for (int i = 0; i < d.P; i++) compact[i] = (float)(i + 1);
        rlc_pack_blob(d, compact.data(), padded.data());
        rlc_unpack_blob(d, padded.data(), back.data());
        for (int i = 0; i < d.P; i++)
            if (back[i] != compact[i]) { printf("round trip failed at %d (blocked %d)\n", i, blocked); return 1; }
        // every device slot is hit at most once and everything else is zero padding
        long nonzero = 0;
        for (int i = 0; i < d.Ppad; i++) nonzero += padded[i] != 0.0f;
        if (nonzero != d.P) { printf("pack wrote %ld slots for %d parameters\n", nonzero, d.P); return 1; }
        if (blocked) {
            // trunk padding rows H1..arow0-1 of Wc2 are zero; the action rows sit at arow0 + j
            for (int r = H1; r < d.arow0; r++)
                for (int c = 0; c < HC; c++)
                    if (padded[d.oWc2 + rlc_blk_index(r, c, HC)] != 0.0f) { printf("padding row %d not zero\n", r); return 1; }
            for (int j = 0; j < A; j++)
                for (int c = 0; c < HC; c++)
                    if (padded[d.oWc2 + rlc_blk_index(d.arow0 + j, c, HC)] != compact[d.seg_compact[6] + (H1 + j) * HC + c]) {
                        printf("action row %d misplaced\n", j);
                        return 1;
                    }
            // lane-contiguity: element (row 16t+c, col 16u+4g+r) of a block sits at ((g*16+c)*4+r) inside its 1 KB
            for (int c = 0; c < 16; c++)
                for (int g = 0; g < 4; g++)
                    for (int r = 0; r < 4; r++)
                        if ((rlc_blk_index(16 + c, 32 + 4 * g + r, HA) & 255) != ((g * 16 + c) * 4 + r)) { printf("lane order\n"); return 1; }
        }
    }
    return 0;
}

int main() {
    const int shapes[][6] = {{3, 1, 200, 200, 200, 100}, {8, 2, 200, 200, 200, 64}, {8, 2, 64, 48, 40, 17},
                             {1, 1, 16, 16, 16, 5}, {3, 1, 128, 128, 128, 128}, {5, 2, 200, 120, 56, 32}};
    for (auto& s : shapes)
        if (check_dims(s[0], s[1], s[2], s[3], s[4], s[5])) return 1;
    // bijection of the block index on a ragged matrix
    const int R = 201, C = 200;
    std::vector<int> seen(rlc_blk_floats(R, C), 0);
    for (int r = 0; r < R; r++)
        for (int c = 0; c < C; c++) {
            const int p = rlc_blk_index(r, c, C);
            if (p < 0 || p >= (int)seen.size() || seen[p]++) { printf("index collision\n"); return 1; }
        }
    printf("OK\n");
    return 0;
}
