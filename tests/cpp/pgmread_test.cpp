// prints "w h v0 v1 ..." of the image readPGMfile() returns; exit code 1 if it returns 0
#include <cstdio>

#include "pgmread.h"

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    int            w = 0, h = 0;
    unsigned char* p = readPGMfile(argv[1], w, h);
    if (!p) return 1;
    printf("%d %d", w, h);
    for (int i = 0; i < w * h; i++) printf(" %d", (int)p[i]);
    printf("\n");
    delete[] p;
    return 0;
}
