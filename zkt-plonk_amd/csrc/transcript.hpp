// Host-side Fiat-Shamir transcripts (the north star keeps Fiat-Shamir on the host).
//
// * Merlin (merlin 3.0: STROBE-128 over Keccak-f[1600], protocol label "Merlin v1.0") wrapped the
//   way plonk-core/src/transcript.rs:46-109 wraps it (MerlinTranscript).
// * EthereumTranscript of gadgets/src/transcript.rs:8-90 (Keccak-256, BN254 only).
// Both sit behind the zkt_transcript callback table of include/zkt_plonk.h, which is also what a
// Rust caller implements on top of its own `T: TranscriptProtocol`.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>

namespace zkt {

void keccak_f1600(uint8_t state[200]);
void keccak256(const uint8_t* data, size_t len, uint8_t out[32]);

class Strobe128 {
  public:
    explicit Strobe128(const std::string& protocol_label);
    void meta_ad(const uint8_t* data, size_t len, bool more);
    void ad(const uint8_t* data, size_t len, bool more);
    void prf(uint8_t* out, size_t len, bool more);

  private:
    static constexpr int R = 166;
    void run_f();
    void absorb(const uint8_t* data, size_t len);
    void squeeze(uint8_t* out, size_t len);
    void begin_op(uint8_t flags, bool more);
    uint8_t st_[200];
    uint8_t pos_ = 0, pos_begin_ = 0, cur_flags_ = 0;
};

class Merlin {
  public:
    explicit Merlin(const std::string& label);
    void append_message(const std::string& label, const uint8_t* msg, size_t len);
    void append_u64(const std::string& label, uint64_t x);
    void challenge_bytes(const std::string& label, uint8_t* out, size_t len);

  private:
    Strobe128 strobe_;
};

// Byte-level transcript interface used by the prover: everything is already serialised the way the
// reference's TranscriptProtocol impls see it (canonical field elements, affine coordinates).
struct HostTranscript {
    virtual ~HostTranscript() {}
    virtual void append_u64(const char* label, uint64_t v) = 0;
    // scalars: count canonical little-endian values of `fr_bytes` bytes each, back to back
    virtual void append_scalars(const char* label, const uint8_t* le, size_t count, size_t fr_bytes, bool single) = 0;
    // affine point: canonical x, y little-endian (fq_bytes each) + infinity flag
    virtual void append_commitment(const char* label, const uint8_t* x_le, const uint8_t* y_le, size_t fq_bytes,
                                   bool infinity) = 0;
    // 32 bytes little-endian canonical challenge
    virtual void challenge_scalar(const char* label, size_t fr_bits, uint8_t out_le[32]) = 0;
};

struct MerlinHostTranscript : HostTranscript {
    explicit MerlinHostTranscript(const std::string& label) : t(label) {}
    void append_u64(const char* label, uint64_t v) override;
    void append_scalars(const char* label, const uint8_t* le, size_t count, size_t fr_bytes, bool single) override;
    void append_commitment(const char* label, const uint8_t* x_le, const uint8_t* y_le, size_t fq_bytes,
                           bool infinity) override;
    void challenge_scalar(const char* label, size_t fr_bits, uint8_t out_le[32]) override;
    Merlin t;
};

struct EthereumHostTranscript : HostTranscript {
    EthereumHostTranscript();
    void append_u64(const char* label, uint64_t v) override;
    void append_scalars(const char* label, const uint8_t* le, size_t count, size_t fr_bytes, bool single) override;
    void append_commitment(const char* label, const uint8_t* x_le, const uint8_t* y_le, size_t fq_bytes,
                           bool infinity) override;
    void challenge_scalar(const char* label, size_t fr_bits, uint8_t out_le[32]) override;

  private:
    void append_bytes(const uint8_t* item, size_t len);
    uint8_t state0_[32], state1_[32];
    uint32_t counter_ = 0;
};

}  // namespace zkt

#include <memory>
// the opaque handle of include/zkt_plonk.h
struct zkt_transcript {
    std::unique_ptr<zkt::HostTranscript> impl;
    zkt::Merlin* merlin = nullptr;  // raw access for the conformance KAT
};
