// Exercises gofindthem_amd/csrc/copy_pool.hpp (the copy threads of the host-memory entry points) on its own: sizes around the
// single-thread cut and the part borders, pools of 0 / 1 / 3 / 11 workers, thousands of generations back to back (a lost
// wake-up would hang, a stale generation would copy the wrong bytes).  Built and run by tests/test_copy_pool.py, under
// ThreadSanitizer where the toolchain has it.
#include "copy_pool.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

int main() {
    for (unsigned w : {0u, 1u, 3u, 11u}) {
        gft::CopyPool p(w);
        if (p.threads() < 1 || p.threads() > w + 1) { puts("thread count"); return 1; }
        for (size_t n : {(size_t)0, (size_t)5, (size_t)(1u << 20) - 1, (size_t)(1u << 20), (size_t)(3u << 20) + 17, (size_t)(16u << 20)}) {
            std::vector<uint8_t> a(n), b(n, 0xEE);
            for (size_t i = 0; i < n; i++) a[i] = (uint8_t)(i * 131 + (i >> 8));
            for (int rep = 0; rep < 3; rep++) {
                std::fill(b.begin(), b.end(), 0xEE);
                p.copy(b.data(), a.data(), n);
                if (n && memcmp(a.data(), b.data(), n)) { printf("MISMATCH workers=%u n=%zu\n", w, n); return 1; }
            }
        }
        std::vector<uint8_t> a(2u << 20, 7), b(2u << 20);
        for (int rep = 0; rep < 300; rep++) {
            a[rep] = (uint8_t)(rep + 1);
            p.copy(b.data(), a.data(), a.size());
            if (b[rep] != (uint8_t)(rep + 1)) { puts("stale generation"); return 1; }
        }
    }
    puts("copy pool ok");
    return 0;
}
