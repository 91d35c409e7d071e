// oracle/ref_hufftable.cpp -- fixture generator (test infrastructure only).
//
// Builds the Huffman code table the REFERENCE would build for a file, by calling
// the reference's own BuildTree()/GenerateCodes() (huffTree.h:55-94, included
// from where it lies via -I/root/reference) on the file's full byte histogram,
// then flattening vector<bool> codes to (codeword, length) pairs the way
// load_data.h:40-47 does (most significant code bit first).  The reference's
// driver itself (load_data.h / hist.cu) needs nvcc and does not compile, so the
// histogram (a plain byte count; SURVEY.md section 5 explains why it equals
// hist.cu's windowed count on the shipped file) and the flattening loop are
// this file's own.
//
// usage: ref_hufftable <input file> <out table blob>
//   blob = codewords[256] (uint32 LE) followed by codewordlens[256] (uint32 LE)
#include <cstdint>
#include <cstdio>
#include <vector>
#include "huffTree.h"

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: %s <input> <table.bin>\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 1; }
    unsigned int freqs[UniqueSymbols] = {0};
    std::vector<unsigned char> buf(1 << 20);
    size_t n;
    while ((n = std::fread(buf.data(), 1, buf.size(), f)) > 0)
        for (size_t i = 0; i < n; ++i) freqs[buf[i]]++;
    std::fclose(f);

    INode* root = BuildTree(freqs);
    HuffCodeMap codes;
    GenerateCodes(root, HuffCode(), codes);
    delete root;

    uint32_t table[512] = {0};
    for (HuffCodeMap::const_iterator it = codes.begin(); it != codes.end(); ++it) {
        const HuffCode& bits = it->second;
        uint32_t cw = 0;
        for (size_t i = 0; i < bits.size(); ++i) cw = (cw << 1) | (bits[i] ? 1u : 0u);
        table[it->first]       = cw;
        table[256 + it->first] = (uint32_t)bits.size();
    }
    FILE* o = std::fopen(argv[2], "wb");
    if (!o) { std::perror(argv[2]); return 1; }
    std::fwrite(table, sizeof(uint32_t), 512, o);
    std::fclose(o);
    return 0;
}
