/* PGM / PPM reader of the demo programs (replaces src/application/pgmread.h). */
#pragma once
#include <string>

/* Reads a P2 / P3 / P5 / P6 file and returns a new[]-allocated w*h 8-bit grayscale image
 * (caller delete[]s), or 0 after printing a message on cerr. */
unsigned char* readPGMfile(const std::string& filename, int& w, int& h);
