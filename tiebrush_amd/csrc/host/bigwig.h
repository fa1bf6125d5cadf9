// bigwig.h — bigWig writer for tiecov -W (the reference hands its intervals to libBigWig: tiecov.cpp:243-275, :365-402).
// Written from the bigWig / bbi file format (Kent et al. 2010, supplement): header, zoom headers, total summary, chromosome
// B+ tree, bedGraph-type data sections (zlib-compressed, up to 1024 items, one chromosome each), R-tree index, and zoom levels
// (32-byte summary records with their own R-trees).  Values are float32, as in the reference's bwAddIntervals call.
// libBigWig is not in this image, so byte identity with the reference's files is not claimed: the intervals a reader gets
// back are the bedGraph lines of `tiecov -c` (tests/test_gpu_cli.py reads the file back with an independent parser).
#pragma once
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

namespace tbh {

class BigWigWriter {
 public:
  // chromosome names and lengths in id order (the BAM header's @SQ order)
  bool open(const std::string& path, const std::vector<std::string>& names, const std::vector<uint32_t>& lens, std::string& err);
  // intervals in (chromosome id, start) order, half-open, 0-based, non-overlapping
  void add(uint32_t chrom, uint32_t start, uint32_t end, float value);
  bool close(std::string& err);

 private:
  struct Item {
    uint32_t chrom, start, end;
    float val;
  };
  struct Block {  // one data (or zoom) section on disk
    uint32_t chrom, start, end;
    uint64_t off, size;
  };
  bool write_index(const std::vector<Block>& blocks, uint64_t* index_off);
  bool write_sections(const std::vector<uint8_t>& payload, uint32_t chrom, uint32_t start, uint32_t end, std::vector<Block>& blocks);
  FILE* f_ = nullptr;
  std::string path_;
  std::vector<std::string> names_;
  std::vector<uint32_t> lens_;
  std::vector<Item> items_;
  uint32_t max_uncompressed_ = 0;
};

}  // namespace tbh
