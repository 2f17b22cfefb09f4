// DEFLATE for the rows of <species>.snps.gz, written for what those rows are.
//
// The reference writes the table through Python's gzip (midas/utility.py:194-206, midas/run/snps.py:179-210); once the
// pileup itself takes a millisecond, zlib's hash chains over ~28 bytes a site are the largest item of the stage
// (8-10 core-seconds for a 15 Mb genome).  A row is
//     <ref_id> \t <ref_pos> \t <ref_allele> \t <depth> \t <count_a> \t <count_c> \t <count_g> \t <count_t> \n
// and its redundancy has two sources the formatter already knows the position of: the head of the row (ref_id and the
// leading digits of ref_pos) repeats the row before it, and the tail from the tab before ref_allele -- allele, depth,
// four counts -- repeats some recent row with the same tuple (a genome at 20x has a few hundred distinct tuples).  So
// the parser does ONE table lookup per row, keyed by the tail, and extends the match through the newline into the next
// row's head; what is left between two matches is a digit or two of ref_pos, sent as literals.  No hash chains, no
// per-byte work beyond the compares.  The tokens go out as one dynamic-Huffman block per gzip member (RFC 1951), so any
// gzip reader inflates the result; on the synthetic 20x tables it is about the size zlib level 4 produces at roughly a
// tenth of the time.  Rows that match nothing (high-depth or polymorphic sites) cost literals, never correctness.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace midas {

class RowDeflate {
 public:
  RowDeflate();
  // Text of one member: `text[0, n)`, rows given by where each one starts and where its tail starts (the tab before
  // ref_allele); row k ends where row k + 1 starts, the last one at n.  Appends a raw DEFLATE stream (BFINAL set) to out.
  void compress(const uint8_t* text, size_t n, const uint32_t* row_begin, const uint32_t* tail_begin, size_t n_rows,
                std::vector<uint8_t>& out);

 private:
  struct Code { uint16_t bits; uint8_t len; };
  void literal(uint8_t b) { tok_ll_.push_back(b); tok_d_.push_back(0); ++freq_ll_[b]; }
  void match(size_t len, size_t dist);
  void build_lengths(const uint32_t* freq, int n, int max_len, uint8_t* len_out);
  static void make_codes(const uint8_t* len, int n, Code* codes);
  void put(uint32_t bits, int n) {          // n <= 28: at most 59 bits are pending before a flush
    acc_ |= (uint64_t)bits << fill_;
    fill_ += n;
    if (fill_ >= 32) {
      const uint32_t w = (uint32_t)acc_;
      __builtin_memcpy(at_, &w, 4);
      at_ += 4;
      acc_ >>= 32;
      fill_ -= 32;
    }
  }
  std::vector<uint16_t> tok_ll_, tok_d_;     // literal byte or match length | match distance (0 = literal)
  uint32_t freq_ll_[288], freq_d_[32];
  std::vector<uint32_t> table_;              // tail hash -> offset + 1 of the latest row tail with that hash
  uint8_t* at_ = nullptr;                    // next output byte
  uint64_t acc_ = 0;
  int fill_ = 0;
};

}  // namespace midas
