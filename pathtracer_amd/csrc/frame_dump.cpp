/* Frame-dump interchange and image files (host side only, no GPU work).
 *
 * The reference dumps the RGBA16F "current image" of a frame as
 *     RayStackBuffer_<index>.bin.z      one zlib stream (miniz mz_compress2, default level) of W*H*8 bytes
 *     RayStackBuffer_<index>.bin.z.txt  "W,H\n"
 * (PathTracer.cpp:180-202, size at :848).  The stream format is RFC 1950, so zlib writes what miniz reads and
 * the other way round; the compressed bytes themselves are not comparable, the inflated texels are.
 */
#include "../../include/ptcore.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

std::string dump_path(const char* dir, uint32_t index)
{
    std::string p = (dir && *dir) ? std::string(dir) : std::string(".");
    if (p.back() != '/') p += '/';
    return p + "RayStackBuffer_" + std::to_string(index) + ".bin.z";              /* PathTracer.cpp:181 */
}

bool write_all(const std::string& path, const void* data, size_t bytes)
{
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = bytes == 0 || fwrite(data, 1, bytes, f) == bytes;
    return (fclose(f) == 0) && ok;
}

bool read_all(const std::string& path, std::vector<unsigned char>& out)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    bool ok = !ferror(f);
    fclose(f);
    return ok;
}

}  // namespace

extern "C" {

/* Function try blocks: an allocation failure comes back as -8, nothing unwinds through the C boundary. */
PT_API int pt_frame_dump_write(const char* dir, uint32_t index, uint32_t width, uint32_t height, const uint16_t* rgba16f)
try {
    if (!rgba16f || !width || !height) return -1;
    const uint64_t bytes = (uint64_t)width * height * 8;
    if (bytes > 0xFFFFFFFFull) return -1;                                        /* the reference's size is a uint32 (PathTracer.cpp:848) */
    uLongf bound = compressBound((uLong)bytes);                                  /* :183 */
    std::vector<unsigned char> z(bound);
    if (compress2(z.data(), &bound, (const Bytef*)rgba16f, (uLong)bytes, Z_DEFAULT_COMPRESSION) != Z_OK) return -5;   /* :186 */
    const std::string path = dump_path(dir, index);
    if (!write_all(path, z.data(), bound)) return -4;                            /* :193-196 */
    char txt[64];
    int n = snprintf(txt, sizeof txt, "%u,%u\n", width, height);                 /* :198-202 */
    if (!write_all(path + ".txt", txt, (size_t)n)) return -4;
    return 0;
} catch (...) { return -8; }

PT_API int pt_frame_dump_read(const char* dir, uint32_t index, uint32_t* width, uint32_t* height, uint16_t* rgba16f,
                              uint64_t capacity_texels)
try {
    if (!width || !height) return -1;
    const std::string path = dump_path(dir, index);
    std::vector<unsigned char> txt;
    if (!read_all(path + ".txt", txt)) return -4;
    txt.push_back(0);
    unsigned w = 0, h = 0;
    if (sscanf((const char*)txt.data(), "%u ,%u", &w, &h) != 2 || !w || !h) return -6;
    *width = w; *height = h;
    if (!rgba16f) return 0;                                                      /* size query */
    const uint64_t texels = (uint64_t)w * h;
    if (capacity_texels < texels) return -7;
    std::vector<unsigned char> z;
    if (!read_all(path, z)) return -4;
    uLongf got = (uLongf)(texels * 8);
    int rc = uncompress((Bytef*)rgba16f, &got, z.data(), (uLong)z.size());
    if (rc != Z_OK) return -5;
    if ((uint64_t)got != texels * 8) return -6;
    return 0;
} catch (...) { return -8; }

PT_API int pt_write_ppm(const char* path, uint32_t width, uint32_t height, const uint8_t* rgba8)
try {
    if (!path || !rgba8 || !width || !height) return -1;
    std::vector<unsigned char> row((size_t)width * 3);
    FILE* f = fopen(path, "wb");
    if (!f) return -4;
    fprintf(f, "P6\n%u %u\n255\n", width, height);
    bool ok = true;
    for (uint32_t y = 0; y < height && ok; y++) {
        const uint8_t* s = rgba8 + (size_t)y * width * 4;
        for (uint32_t x = 0; x < width; x++) { row[3 * x] = s[4 * x]; row[3 * x + 1] = s[4 * x + 1]; row[3 * x + 2] = s[4 * x + 2]; }
        ok = fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    return (fclose(f) == 0 && ok) ? 0 : -4;
} catch (...) { return -8; }

}  // extern "C"
