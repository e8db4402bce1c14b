// The placement arithmetic of rt_render_multi / place_columns_kernel on the CPU: for every width, height and number of
// shares tried, the shares' compact elements land on every frame element exactly once, in column order.
#include <cstdio>
#include <vector>

#include "rt_partition.h"

int main()
{
    const int widths[] = { 1, 2, 5, 7, 64, 101, 800, 3840 }, heights[] = { 1, 3, 64 }, shares[] = { 1, 2, 3, 4, 7, 8, 16 };
    long long checked = 0;
    for (int W : widths)
        for (int H : heights)
            for (int n : shares) {
                std::vector<int> seen((size_t)W * H, 0);
                int total_cols = 0;
                for (int r = 0; r < n; r++) {
                    const int nc = rt::share_columns(W, r, n);
                    total_cols += nc;
                    for (int k = 0; k < nc; k++)
                        for (int py = 0; py < H; py++) {
                            const long long d = rt::frame_index(r, n, k, H, py);
                            if (d < 0 || d >= (long long)W * H) { std::printf("out of range W=%d H=%d n=%d r=%d k=%d\n", W, H, n, r, k); return 1; }
                            if (d != (long long)(r + k * n) * H + py || (int)(d / H) % n != r) { std::printf("wrong column W=%d n=%d r=%d k=%d\n", W, n, r, k); return 1; }
                            seen[(size_t)d]++;
                            checked++;
                        }
                }
                if (total_cols != W) { std::printf("columns W=%d n=%d: %d\n", W, n, total_cols); return 1; }
                for (int v : seen) if (v != 1) { std::printf("coverage W=%d H=%d n=%d\n", W, H, n); return 1; }
            }
    std::printf("ok %lld\n", checked);
    return 0;
}
