// sam.h — SAM text input (sam.cpp): probe and conversion into the uncompressed BAM byte stream BamFile indexes.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace tbh {
// the first line is a SAM header line or has the eleven mandatory fields of an alignment line
bool sam_probe(const std::string& path);
// the file starts with the CRAM magic
bool cram_probe(const std::string& path);
// whole file -> "BAM\1" + header block + records (no BGZF framing); false + err on malformed input
bool sam_to_bam(const std::string& path, std::vector<uint8_t>& out, std::string& err);
}  // namespace tbh
