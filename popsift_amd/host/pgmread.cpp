/*
 * PGM / PPM reader of the demo programs (replaces src/application/pgmread.cpp:38-253).
 *
 * Header parsing follows the Netpbm grammar (magic, width, height, maxval as whitespace-separated
 * tokens, '#' comments to end of line, ONE whitespace byte before binary rasters), which accepts
 * every file the reference's line-based parser accepts.  Sample conversion is the reference's,
 * so the same file gives the same bytes:
 *   - plain (P2/P3) samples: value if maxval == 255, else (unsigned char)(value * 255.0 / maxval)
 *   - colour -> gray in integers with OpenCV's weights: (4899 r + 9617 g + 1868 b) >> 14
 *   - 16-bit binary samples are read in HOST byte order (pgmread.cpp:181-192 reads the raw
 *     shorts; Netpbm says big-endian -- kept as is for identical results) and, for P6, are NOT
 *     rescaled before the gray conversion (pgmread.cpp:226-246)
 */
#include "pgmread.h"

#include <cctype>
#include <cstdint>
#include <fstream>
#include <iostream>
#include <sys/stat.h>
#include <vector>

using namespace std;

namespace {

/* next header token; skips whitespace and comments; leaves the stream right after the token */
bool header_token(istream& in, string& tok)
{
    tok.clear();
    int c;
    for (;;) {
        c = in.get();
        if (c == EOF) return false;
        if (c == '#') {
            while (c != EOF && c != '\n') c = in.get();
            continue;
        }
        if (!isspace(c)) break;
    }
    while (c != EOF && !isspace(c)) {
        tok.push_back((char)c);
        c = in.get();
    }
    if (c != EOF) in.unget(); /* the single separator byte is consumed by the caller */
    return true;
}

bool header_int(istream& in, int& v)
{
    string t;
    if (!header_token(in, t)) return false;
    char* end = 0;
    const long r = strtol(t.c_str(), &end, 10);
    if (end == t.c_str() || *end != 0) return false;
    v = (int)r;
    return true;
}

inline unsigned char plain_sample(int input, int maxval)
{
    return (maxval == 255) ? (unsigned char)input : (unsigned char)(input * 255.0 / maxval);
}

inline unsigned char gray(unsigned int r, unsigned int g, unsigned int b)
{
    return (unsigned char)((4899u * r + 9617u * g + 1868u * b) >> 14);
}

}  // namespace

unsigned char* readPGMfile(const string& filename, int& w, int& h)
{
    struct stat st;
    if (stat(filename.c_str(), &st) != 0) {
        cerr << "File \"" << filename << "\" does not exist" << endl;
        return 0;
    }
    ifstream f(filename.c_str(), ios::binary);
    if (!f.is_open()) {
        cerr << "File \"" << filename << "\" could not be opened for reading" << endl;
        return 0;
    }
    string magic;
    if (!header_token(f, magic)) {
        cerr << "File \"" << filename << "\" is too short" << endl;
        return 0;
    }
    int type = 0;
    if (magic.size() >= 2 && magic[0] == 'P' && (magic[1] == '2' || magic[1] == '3' || magic[1] == '5' || magic[1] == '6'))
        type = magic[1] - '0';
    if (type == 0) {
        cerr << "File \"" << filename << "\" can only contain P2, P3, P5 or P6 PGM images" << endl;
        return 0;
    }
    int maxval = 0;
    if (!header_int(f, w) || !header_int(f, h)) {
        cerr << "File \"" << filename << "\" PGM type header (" << type << ") must be followed by comments and WxH info"
             << endl;
        return 0;
    }
    if (w <= 0 || h <= 0) {
        cerr << "File \"" << filename << "\" has meaningless image size" << endl;
        return 0;
    }
    if (!header_int(f, maxval)) {
        cerr << "File \"" << filename << "\" PGM dimensions must be followed by comments and max value info" << endl;
        return 0;
    }
    f.get(); /* the one whitespace byte that ends the header */

    const size_t   n = (size_t)w * (size_t)h;
    unsigned char* out = new unsigned char[n];
    auto           too_short = [&]() -> unsigned char* {
        cerr << "File \"" << filename << "\" file too short" << endl;
        delete[] out;
        return 0;
    };

    if (type == 2) {
        for (size_t i = 0; i < n; i++) {
            int v;
            f >> v;
            if (f.fail()) return too_short();
            out[i] = plain_sample(v, maxval);
        }
    } else if (type == 3) {
        for (size_t i = 0; i < n; i++) {
            int rgb[3];
            f >> rgb[0] >> rgb[1] >> rgb[2];
            if (f.fail()) return too_short();
            out[i] = gray(plain_sample(rgb[0], maxval), plain_sample(rgb[1], maxval), plain_sample(rgb[2], maxval));
        }
    } else if (type == 5) {
        if (maxval < 256) {
            f.read((char*)out, (streamsize)n); /* a short raster is not an error in the reference either */
        } else {
            vector<uint16_t> raw(n);
            f.read((char*)raw.data(), (streamsize)(n * 2));
            if (f.fail()) return too_short();
            for (size_t i = 0; i < n; i++) out[i] = (unsigned char)(raw[i] * 255.0 / maxval);
        }
    } else { /* P6 */
        if (maxval < 256) {
            vector<unsigned char> raw(n * 3);
            f.read((char*)raw.data(), (streamsize)(n * 3));
            if (f.fail()) return too_short();
            for (size_t i = 0; i < n; i++) out[i] = gray(raw[3 * i], raw[3 * i + 1], raw[3 * i + 2]);
        } else {
            vector<uint16_t> raw(n * 3);
            f.read((char*)raw.data(), (streamsize)(n * 6));
            if (f.fail()) return too_short();
            for (size_t i = 0; i < n; i++) out[i] = gray(raw[3 * i], raw[3 * i + 1], raw[3 * i + 2]);
        }
    }
    return out;
}
