// Readers for the key files the reference CLI writes: `serialize_to_file` =
// CanonicalSerialize::serialize_unchecked (bin/src/parser.rs:14-22) of
//   * the SonicKZG10 CommitterKey  (bin/src/main.rs:105 --ck)   -> powers_of_g           -> zkt_srs_load
//   * plonk-core's ProverKey<F>    (bin/src/main.rs:107 --pk; keys/mod.rs:29-41)         -> zkt_circuit_load
//   * plonk-core's VerifierKey     (bin/src/main.rs:111 --vk; keys/mod.rs:180-210)       -> zkt_transcript_seed
//   * plonk-core's ExtendedProverKey<F> (bin/src/main.rs:34-35,108-109 --epk; keys/mod.rs:148-174): never NEEDED -- the
//     extended key is derived on the device by zkt_circuit_load -- but read, vector by vector, so that a file the reference
//     wrote can be checked against that derivation (zkt_circuit_check_epk_file, prover.hip)
// Host-only code.  The byte layouts follow ark-serialize 0.3 / ark-poly-commit 0.3 (third-party crates absent from
// /root/reference; restated from their published derive rules, no reference-held file exists to pin them against:
// "parity unpinned", the round trip is tested against the writer of the test-side restatement):
//   usize, u64                -> 8 bytes little endian          Vec<T> -> u64 length, then the elements
//   String                    -> Vec<u8>                        Option<T> -> one byte (0 / 1), then T
//   Fp256 / Fp384             -> canonical (non-Montgomery) value, little endian, 32 / 48 bytes
//   GroupAffine, unchecked    -> x, then y with the SW flags in the top bits of its last byte (bit 6 = infinity)
//   DensePolynomial           -> coeffs: Vec<F>
//   LabeledPolynomial         -> label: String, polynomial, degree_bound: Option<usize>, hiding_bound: Option<usize>
//   sonic_pc::CommitterKey    -> powers_of_g: Vec<G1Affine> first (the only field the prover needs)
#include "ctx.hpp"
#include "ec.hpp"
#include "keyfile.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace zkt {

int srs_load(zkt_ctx* c, const void* src, size_t count, bool on_device, size_t slice_off, size_t total);   // msm.hip

struct Cursor {
    const uint8_t* p;
    size_t len, pos = 0;
    bool ok = true;
    bool need(size_t k) {
        if (!ok || k > len - pos) {
            ok = false;
            return false;
        }
        return true;
    }
    uint64_t u64() {
        if (!need(8)) return 0;
        uint64_t v = 0;
        for (int i = 0; i < 8; ++i) v |= (uint64_t)p[pos + i] << (8 * i);
        pos += 8;
        return v;
    }
    uint8_t u8() {
        if (!need(1)) return 0;
        return p[pos++];
    }
    bool skip(size_t k) {
        if (!need(k)) return false;
        pos += k;
        return true;
    }
    bool option_usize() {   // Option<usize>
        const uint8_t tag = u8();
        if (tag > 1) ok = false;
        if (tag == 1) (void)u64();
        return ok;
    }
};

static bool read_file(const char* path, std::vector<uint8_t>& out, size_t max_bytes = 0) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    if (fseek(f, 0, SEEK_END) != 0) {
        fclose(f);
        return false;
    }
    long sz = ftell(f);
    if (sz < 0) {
        fclose(f);
        return false;
    }
    size_t want = (size_t)sz;
    if (max_bytes && want > max_bytes) want = max_bytes;   // a CommitterKey holds 4n + 1 powers; the prover takes n + 8
    rewind(f);
    out.resize(want);
    const size_t got = want ? fread(out.data(), 1, want, f) : 0;
    fclose(f);
    return got == want;
}

// canonical little-endian bytes -> arkworks Montgomery limbs; false when the value is not below the modulus
template <class P>
static bool field_from_bytes(const uint8_t* b, uint32_t strip_top_bits, uint64_t* out) {
    Fe<P> v;
    memcpy(v.v, b, P::N * 4);
    if (strip_top_bits) v.v[P::N - 1] &= (0xFFFFFFFFu >> strip_top_bits);
    for (int i = P::N - 1; i >= 0; --i) {
        if (v.v[i] < P::mod(i)) break;
        if (v.v[i] > P::mod(i)) return false;
        if (i == 0) return false;   // equal to the modulus
    }
    v = fe_to_mont<P>(v);
    memcpy(out, v.v, P::N * 4);
    return true;
}

// one unchecked G1Affine -> x || y Montgomery limbs, (0, 0) for the point at infinity
template <class Q>
static bool point_from_bytes(Cursor& cur, uint64_t* out_xy, int* out_inf) {
    const size_t nb = Q::N * 4;
    if (!cur.need(2 * nb)) return false;
    const uint8_t* x = cur.p + cur.pos;
    const uint8_t* y = x + nb;
    cur.pos += 2 * nb;
    const uint8_t flags = y[nb - 1];
    const bool inf = (flags & 0x40) != 0;
    if (out_inf) *out_inf = inf ? 1 : 0;
    if (inf) {
        memset(out_xy, 0, 2 * nb);
        return true;
    }
    // SWFlags live in the two top bits of y's last byte; both moduli leave them free
    return field_from_bytes<Q>(x, 0, out_xy) && field_from_bytes<Q>(y, 2, out_xy + Q::N / 2);
}

template <class C>
static int committer_key_t(const std::vector<uint8_t>& buf, bool truncated, size_t max_powers, uint64_t* out, size_t* n_powers) {
    using Q = typename C::Fq;
    Cursor cur{buf.data(), buf.size()};
    const uint64_t count = cur.u64();
    if (!cur.ok) return ZKT_ERR_INVALID_ARGUMENT;
    const size_t nb = 2 * Q::N * 4;
    if (!truncated && count > (buf.size() - 8) / nb) return ZKT_ERR_INVALID_ARGUMENT;   // length beyond the file
    size_t take = (size_t)count;
    if (max_powers && take > max_powers) take = max_powers;
    if (take > (buf.size() - 8) / nb) return ZKT_ERR_INVALID_ARGUMENT;
    *n_powers = take;
    if (!out) return ZKT_OK;
    for (size_t i = 0; i < take; ++i)
        if (!point_from_bytes<Q>(cur, out + i * 2 * (Q::N / 2), nullptr)) return ZKT_ERR_INVALID_ARGUMENT;
    return ZKT_OK;
}

template <class C>
static int prover_key_t(const std::vector<uint8_t>& buf, uint64_t* const* out_polys, size_t* lens) {
    using R = typename C::Fr;
    Cursor cur{buf.data(), buf.size()};
    // arith { q_m q_l q_r q_o q_c }, perm { sigma1 sigma2 sigma3 }, lookup { q_lookup q_table }: the order of
    // zkt_circuit_load (keys/mod.rs:29-41, keys/arithmetic.rs:20-32, keys/permutation.rs:20-31, keys/lookup.rs:19-25)
    for (int k = 0; k < 10; ++k) {
        const uint64_t label_len = cur.u64();
        if (!cur.ok || label_len > 256 || !cur.skip((size_t)label_len)) return ZKT_ERR_INVALID_ARGUMENT;
        const uint64_t n = cur.u64();
        if (!cur.ok || n > (cur.len - cur.pos) / 32) return ZKT_ERR_INVALID_ARGUMENT;
        lens[k] = (size_t)n;
        if (out_polys && out_polys[k]) {
            for (size_t i = 0; i < (size_t)n; ++i)
                if (!field_from_bytes<R>(cur.p + cur.pos + 32 * i, 0, out_polys[k] + 4 * i)) return ZKT_ERR_INVALID_ARGUMENT;
        }
        cur.skip((size_t)n * 32);
        if (!cur.option_usize() || !cur.option_usize()) return ZKT_ERR_INVALID_ARGUMENT;   // degree_bound, hiding_bound
    }
    return cur.pos == cur.len ? ZKT_OK : ZKT_ERR_INVALID_ARGUMENT;
}

template <class C>
static int verifier_key_t(const std::vector<uint8_t>& buf, uint64_t* n_out, uint64_t* pi_roots, size_t pi_cap, size_t* n_pi,
                          uint64_t* commits, int* is_inf) {
    using R = typename C::Fr;
    using Q = typename C::Fq;
    Cursor cur{buf.data(), buf.size()};
    const uint64_t n = cur.u64();
    const uint64_t k = cur.u64();
    if (!cur.ok || k > (cur.len - cur.pos) / 32) return ZKT_ERR_INVALID_ARGUMENT;
    *n_out = n;
    *n_pi = (size_t)k;
    if (pi_roots) {
        if (k > pi_cap) return ZKT_ERR_INVALID_ARGUMENT;
        for (size_t i = 0; i < (size_t)k; ++i)
            if (!field_from_bytes<R>(cur.p + cur.pos + 32 * i, 0, pi_roots + 4 * i)) return ZKT_ERR_INVALID_ARGUMENT;
    }
    cur.skip((size_t)k * 32);
    // arith { q_m q_l q_r q_o q_c }, perm { sigma1 sigma2 sigma3 }, lookup { q_lookup q_table } (keys/mod.rs:196-210)
    for (int j = 0; j < 10; ++j) {
        uint64_t xy[12];
        int inf = 0;
        if (!point_from_bytes<Q>(cur, xy, &inf)) return ZKT_ERR_INVALID_ARGUMENT;
        if (commits) memcpy(commits + (size_t)j * 2 * (Q::N / 2), xy, 2 * Q::N * 4);
        if (is_inf) is_inf[j] = inf;
    }
    return cur.pos == cur.len ? ZKT_OK : ZKT_ERR_INVALID_ARGUMENT;
}


// ---- ExtendedProverKey: seventeen Vec<F>, streamed (the file of a 2^20 circuit holds 1.9 GB) -------------------------
bool EpkReader::open(const char* path) {
    f = fopen(path, "rb");
    if (!f) return false;
    if (fseeko(f, 0, SEEK_END) != 0) return false;
    const off_t sz = ftello(f);
    if (sz < 0 || fseeko(f, 0, SEEK_SET) != 0) return false;
    size = (uint64_t)sz;
    pos = 0;
    left = 0;
    return true;
}
void EpkReader::close() {
    if (f) fclose(f);
    f = nullptr;
}
bool EpkReader::next(uint64_t* len) {
    if (!f || left != 0 || size - pos < 8) return false;
    uint8_t b[8];
    if (fread(b, 1, 8, f) != 8) return false;
    pos += 8;
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i) v |= (uint64_t)b[i] << (8 * i);
    if (v > (size - pos) / 32) return false;   // a length beyond the file
    left = v;
    *len = v;
    return true;
}
bool EpkReader::read(uint8_t* dst, size_t elems) {
    if (!f || elems > left) return false;
    if (elems && fread(dst, 32, elems, f) != elems) return false;
    left -= elems;
    pos += (uint64_t)elems * 32;
    return true;
}
bool EpkReader::skip() {
    if (!f) return false;
    if (left && fseeko(f, (off_t)(left * 32), SEEK_CUR) != 0) return false;
    pos += left * 32;
    left = 0;
    return true;
}

template <class R>
static int epk_vector_t(EpkReader& rd, int which, uint64_t* out, size_t cap, size_t* lens) {
    std::vector<uint8_t> chunk;
    for (int k = 0; k < EPK_VECTORS; ++k) {
        uint64_t len = 0;
        if (!rd.next(&len)) return ZKT_ERR_INVALID_ARGUMENT;
        lens[k] = (size_t)len;
        if (k == which && out) {
            if (len > cap) return ZKT_ERR_INVALID_ARGUMENT;
            const size_t step = 1u << 16;
            chunk.resize(step * 32);
            for (size_t i = 0; i < (size_t)len; i += step) {
                const size_t cnt = std::min(step, (size_t)len - i);
                if (!rd.read(chunk.data(), cnt)) return ZKT_ERR_INVALID_ARGUMENT;
                for (size_t j = 0; j < cnt; ++j)
                    if (!field_from_bytes<R>(chunk.data() + 32 * j, 0, out + 4 * (i + j))) return ZKT_ERR_INVALID_ARGUMENT;
            }
        } else if (!rd.skip()) {
            return ZKT_ERR_INVALID_ARGUMENT;
        }
    }
    return rd.at_end() ? ZKT_OK : ZKT_ERR_INVALID_ARGUMENT;   // trailing bytes: not an ExtendedProverKey
}

}  // namespace zkt

using namespace zkt;

extern "C" {

int zkt_keyfile_committer_key(const char* path, int curve_id, size_t max_powers, uint64_t* out_xy_mont, size_t* n_powers) {
    if (!path || !n_powers || (curve_id != ZKT_CURVE_BN254 && curve_id != ZKT_CURVE_BLS12_381)) return ZKT_ERR_INVALID_ARGUMENT;
    const size_t pt = curve_id == ZKT_CURVE_BN254 ? 64 : 96;
    std::vector<uint8_t> buf;
    if (!read_file(path, buf, max_powers ? 8 + max_powers * pt : 0)) return ZKT_ERR_INVALID_ARGUMENT;
    const bool truncated = max_powers != 0;
    if (curve_id == ZKT_CURVE_BN254) return committer_key_t<Bn254Curve>(buf, truncated, max_powers, out_xy_mont, n_powers);
    return committer_key_t<Bls381Curve>(buf, truncated, max_powers, out_xy_mont, n_powers);
}

int zkt_keyfile_prover_key(const char* path, int curve_id, uint64_t* const* out_polys, size_t* lens) {
    if (!path || !lens || (curve_id != ZKT_CURVE_BN254 && curve_id != ZKT_CURVE_BLS12_381)) return ZKT_ERR_INVALID_ARGUMENT;
    std::vector<uint8_t> buf;
    if (!read_file(path, buf)) return ZKT_ERR_INVALID_ARGUMENT;
    if (curve_id == ZKT_CURVE_BN254) return prover_key_t<Bn254Curve>(buf, out_polys, lens);
    return prover_key_t<Bls381Curve>(buf, out_polys, lens);
}

int zkt_keyfile_verifier_key(const char* path, int curve_id, uint64_t* n, uint64_t* pi_roots_mont, size_t pi_cap, size_t* n_pi,
                             uint64_t* commitments_xy_mont, int* is_infinity) {
    if (!path || !n || !n_pi || (curve_id != ZKT_CURVE_BN254 && curve_id != ZKT_CURVE_BLS12_381)) return ZKT_ERR_INVALID_ARGUMENT;
    std::vector<uint8_t> buf;
    if (!read_file(path, buf)) return ZKT_ERR_INVALID_ARGUMENT;
    if (curve_id == ZKT_CURVE_BN254) return verifier_key_t<Bn254Curve>(buf, n, pi_roots_mont, pi_cap, n_pi, commitments_xy_mont, is_infinity);
    return verifier_key_t<Bls381Curve>(buf, n, pi_roots_mont, pi_cap, n_pi, commitments_xy_mont, is_infinity);
}

int zkt_keyfile_extended_prover_key(const char* path, int curve_id, int which, uint64_t* out_mont, size_t cap, size_t* lens17) {
    if (!path || !lens17 || which < -1 || which >= EPK_VECTORS || (curve_id != ZKT_CURVE_BN254 && curve_id != ZKT_CURVE_BLS12_381))
        return ZKT_ERR_INVALID_ARGUMENT;
    EpkReader rd;
    if (!rd.open(path)) {
        rd.close();
        return ZKT_ERR_INVALID_ARGUMENT;
    }
    const int rc = curve_id == ZKT_CURVE_BN254 ? epk_vector_t<Bn254Fr>(rd, which, out_mont, cap, lens17)
                                               : epk_vector_t<Bls381Fr>(rd, which, out_mont, cap, lens17);
    rd.close();
    return rc;
}

int zkt_srs_load_file(zkt_ctx* c, const char* ck_path, size_t max_powers) {
    if (!c || !ck_path) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    size_t count = 0;
    int rc = zkt_keyfile_committer_key(ck_path, c->curve, max_powers, nullptr, &count);
    if (rc || count == 0) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, std::string("cannot read a CommitterKey from ") + ck_path);
    const size_t words = c->curve == ZKT_CURVE_BN254 ? 8 : 12;
    std::vector<uint64_t> pts(count * words);
    if ((rc = zkt_keyfile_committer_key(ck_path, c->curve, max_powers, pts.data(), &count)))
        return set_err(c, rc, std::string("malformed point in ") + ck_path);
    (void)hipSetDevice(c->device);
    return srs_load(c, pts.data(), count, false, 0, 0);
}

int zkt_circuit_load_file(zkt_ctx* c, const char* pk_path, int log_n) {
    if (!c || !pk_path) return set_err(c, ZKT_ERR_INVALID_ARGUMENT, "null pointer");
    size_t lens[10] = {};
    int rc = zkt_keyfile_prover_key(pk_path, c->curve, nullptr, lens);
    if (rc) return set_err(c, rc, std::string("cannot read a ProverKey from ") + pk_path);
    std::vector<std::vector<uint64_t>> polys(10);
    uint64_t* ptrs[10];
    for (int k = 0; k < 10; ++k) {
        polys[k].resize(lens[k] * 4 + 4);
        ptrs[k] = polys[k].data();
    }
    if ((rc = zkt_keyfile_prover_key(pk_path, c->curve, ptrs, lens))) return set_err(c, rc, std::string("malformed coefficient in ") + pk_path);
    return zkt_circuit_load(c, log_n, ptrs, lens);
}

}  // extern "C"
