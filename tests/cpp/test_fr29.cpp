// Host harness around the nine-limb field arithmetic the QAP kernels use (falcon-r1cs_amd/csrc/frw_fr29.h), compiled with
// g++ through tests/cpp/hip_host.  Protocol (tests/test_fr29_host.py): one operation per input line,
//   mul a b | add a b | sub2 a b | sub4 a b | sub8 a b | red4 a | canon a | packunpack a | redc x      (hex integers)
// a, b are given as integers and split into nine 29-bit limbs here (the top limb takes what is left), x into eleven;
// the answer is the result's limbs recombined, in hex.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <iostream>
#include <sstream>
#include "frw_fr29.h"

using namespace frw;
typedef unsigned __int128 u128;

struct Big { std::vector<uint32_t> w; };        // little-endian 32-bit words

static Big parse(const std::string &hex)
{
    Big b;
    b.w.assign(12, 0);
    int bit = 0;
    for (int i = (int)hex.size() - 1; i >= 0; i--, bit += 4) {
        const char c = hex[i];
        const uint32_t v = c <= '9' ? c - '0' : (c | 32) - 'a' + 10;
        b.w[bit >> 5] |= v << (bit & 31);
    }
    return b;
}
static uint32_t bits(const Big &b, int lo, int n)          // n <= 32 bits starting at bit lo
{
    uint64_t two = (uint64_t)b.w[lo >> 5] | ((uint64_t)b.w[(lo >> 5) + 1] << 32);
    return (uint32_t)((two >> (lo & 31)) & (n == 32 ? 0xffffffffull : ((1ull << n) - 1)));
}
static F29 to_f29(const Big &b)
{
    F29 r;
    for (int i = 0; i < 8; i++) r.l[i] = bits(b, 29 * i, 29);
    r.l[8] = bits(b, 232, 32);                             // the top limb takes the rest (values up to 2^264)
    return r;
}
static std::string hex_of(const uint32_t *limbs, int n)
{
    // sum limbs[i] 2^(29 i) as hex
    std::vector<uint32_t> w(16, 0);
    for (int i = 0; i < n; i++) {
        const int bit = 29 * i;
        uint64_t v = (uint64_t)limbs[i] << (bit & 31);
        int k = bit >> 5;
        while (v) {
            const uint64_t s = (uint64_t)w[k] + (uint32_t)v;
            w[k] = (uint32_t)s;
            v = (v >> 32) + (s >> 32);
            k++;
        }
    }
    char buf[16 * 8 + 1];
    int p = 0;
    bool started = false;
    for (int k = 15; k >= 0; k--) {
        if (!started && !w[k] && k) continue;
        p += snprintf(buf + p, sizeof(buf) - p, started ? "%08x" : "%x", w[k]);
        started = true;
    }
    return std::string(buf);
}

int main()
{
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream in(line);
        std::string op, sa, sb;
        in >> op >> sa >> sb;
        if (op.empty()) continue;
        const Big A = parse(sa), B = sb.empty() ? Big{std::vector<uint32_t>(12, 0)} : parse(sb);
        F29 r{};
        if (op == "redc") {
            uint32_t x[11];
            for (int i = 0; i < 10; i++) x[i] = bits(A, 29 * i, 29);
            x[10] = bits(A, 290, 32);
            r = f29_redc_wide(x);
        } else {
            const F29 a = to_f29(A), b = to_f29(B);
            if (op == "mul") r = f29_mul(a, b);
            else if (op == "add") r = f29_add(a, b);
            else if (op == "sub2") r = f29_sub_kp<2>(a, b);
            else if (op == "sub4") r = f29_sub_kp<4>(a, b);
            else if (op == "sub8") r = f29_sub_kp<8>(a, b);
            else if (op == "sub16") r = f29_sub_kp<16>(a, b);
            else if (op == "sub32") r = f29_sub_kp<32>(a, b);
            else if (op == "csub8") r = f29_cond_sub_kp<8>(a);
            else if (op == "red4") r = f29_reduce_4p(a);
            else if (op == "canon") r = f29_canonical(a);
            else if (op == "packunpack") r = f29_unpack(f29_pack(a));
            else { std::printf("?\n"); continue; }
        }
        bool normalised = true;
        for (int i = 0; i < 8; i++) normalised &= r.l[i] < (1u << 29);
        std::printf("%s %d\n", hex_of(r.l, 9).c_str(), normalised ? 1 : 0);
    }
    return 0;
}
