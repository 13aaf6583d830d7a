// tools/png_fuzz.cpp — the PNG decoder of the KITTI runner (libviso_amd/host/png_read.hpp) against hostile input, under
// AddressSanitizer / UBSan on the CPU: the decoder reads files it does not trust, and the round-4 inflate works with
// raw pointers (64-bit refills, word-wide match copies) where the round-3 one pushed bytes into a vector.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all tools/png_fuzz.cpp -o png_fuzz
//   ./png_fuzz seed.png [iterations]
// Every iteration copies the seed file, damages it (random byte flips, truncation, a chunk length rewritten, a run of
// zeros / 0xff inside the IDAT payload) and decodes it; whatever comes back must be "refused" or an image of the header's
// size — never a crash, a sanitizer report or an allocation beyond the header's implied size.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../libviso_amd/host/png_read.hpp"

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: png_fuzz seed.png [iterations]\n"); return 2; }
    std::vector<uint8_t> seed;
    if (!viso::png_detail::read_file(argv[1], seed) || seed.size() < 64) { std::fprintf(stderr, "cannot read the seed\n"); return 2; }
    const int iters = argc > 2 ? std::atoi(argv[2]) : 2000;
    const std::string tmp = std::string(argv[1]) + ".fuzz";
    unsigned long long s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    int rows = 0, cols = 0, ok = 0, refused = 0;
    std::vector<uint8_t> gray;
    if (!viso::read_png_gray(argv[1], rows, cols, gray)) { std::fprintf(stderr, "the seed itself is refused\n"); return 2; }
    const int rows0 = rows, cols0 = cols;
    for (int it = 0; it < iters; ++it) {
        std::vector<uint8_t> f = seed;
        switch (rnd() % 5) {
        case 0: for (int k = 0, n = 1 + (int)(rnd() % 8); k < n; ++k) f[rnd() % f.size()] ^= (uint8_t)(1u << (rnd() % 8)); break;
        case 1: f.resize(8 + rnd() % (f.size() - 8)); break;
        case 2: { const size_t o = 33 + rnd() % (f.size() - 40); for (size_t k = 0; k < 4; ++k) f[o + k] = (uint8_t)rnd(); break; }
        case 3: { const size_t o = 41 + rnd() % (f.size() - 60), n = 1 + rnd() % 64; std::memset(&f[o], (rnd() & 1) ? 0x00 : 0xff, std::min(n, f.size() - o)); break; }
        default: for (int k = 0; k < 64; ++k) f[41 + rnd() % (f.size() - 48)] = (uint8_t)rnd(); break;
        }
        FILE* fp = std::fopen(tmp.c_str(), "wb");
        if (!fp) return 2;
        std::fwrite(f.data(), 1, f.size(), fp);
        std::fclose(fp);
        if (viso::read_png_gray(tmp, rows, cols, gray)) {
            if (gray.size() != (size_t)rows * cols || rows <= 0 || cols <= 0) { std::printf("inconsistent result at iteration %d\n", it); return 1; }
            ++ok;
        } else {
            if (!gray.empty() || rows || cols) { std::printf("a refused file left data behind at iteration %d\n", it); return 1; }
            ++refused;
        }
        // and into caller memory of the seed's geometry: must refuse another size, must not write past rows0 * cols0
        std::vector<uint8_t> dst((size_t)rows0 * cols0 + 16, 0xA5);
        viso::read_png_gray_to(tmp, rows0, cols0, dst.data());
        for (int k = 0; k < 16; ++k) if (dst[(size_t)rows0 * cols0 + k] != 0xA5) { std::printf("wrote past the buffer at iteration %d\n", it); return 1; }
    }
    std::remove(tmp.c_str());
    std::printf("png_fuzz ok: %d iterations, %d decoded, %d refused\n", iters, ok, refused);
    return 0;
}
