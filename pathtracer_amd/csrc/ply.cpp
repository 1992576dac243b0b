/*
 * ply.cpp — PLY reader for the scenes the reference renders.
 *
 * Stands in for HimaliaPlyModel::LoadFile + BuildVertices<VertexAux> + BuildVertices<vec4> +
 * BuildIndices<uint32> (reference PathTracer.cpp:246-262; the loader itself lives in a
 * submodule that is not part of the reference checkout).  Input format as in
 * Models/CBbunny.ply:1-17: `element vertex` with x,y,z and optionally nx,ny,nz and
 * red,green,blue,alpha, `element face` with one list property; `format ascii 1.0` or
 * `format binary_little_endian 1.0`.  Polygons with more than three corners are
 * fan-triangulated around their first corner.  Position w is 1.
 */
#include "../../include/ptcore.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

enum Type { T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64, T_BAD };

Type parse_type(const std::string& s)
{
    if (s == "char" || s == "int8") return T_I8;
    if (s == "uchar" || s == "uint8") return T_U8;
    if (s == "short" || s == "int16") return T_I16;
    if (s == "ushort" || s == "uint16") return T_U16;
    if (s == "int" || s == "int32") return T_I32;
    if (s == "uint" || s == "uint32") return T_U32;
    if (s == "float" || s == "float32") return T_F32;
    if (s == "double" || s == "float64") return T_F64;
    return T_BAD;
}
size_t type_size(Type t)
{
    switch (t) { case T_I8: case T_U8: return 1; case T_I16: case T_U16: return 2; case T_I32: case T_U32: case T_F32: return 4; case T_F64: return 8; default: return 0; }
}
double load_num(const unsigned char* p, Type t)
{
    switch (t) {
        case T_I8: { int8_t v; memcpy(&v, p, 1); return v; }
        case T_U8: return *p;
        case T_I16: { int16_t v; memcpy(&v, p, 2); return v; }
        case T_U16: { uint16_t v; memcpy(&v, p, 2); return v; }
        case T_I32: { int32_t v; memcpy(&v, p, 4); return v; }
        case T_U32: { uint32_t v; memcpy(&v, p, 4); return v; }
        case T_F32: { float v; memcpy(&v, p, 4); return v; }
        case T_F64: { double v; memcpy(&v, p, 8); return v; }
        default: return 0;
    }
}

struct Prop { std::string name; Type type = T_BAD; bool list = false; Type countType = T_BAD; };
struct Element { std::string name; size_t count = 0; std::vector<Prop> props; };

struct Reader {
    FILE* f = nullptr;
    bool ascii = true;
    std::vector<Element> elements;
    std::string error;

    ~Reader() { if (f) fclose(f); }

    bool fail(const std::string& m) { error = m; return false; }

    bool read_line(std::string& out)
    {
        out.clear();
        int c;
        while ((c = fgetc(f)) != EOF) {
            if (c == '\n') return true;
            if (c != '\r') out.push_back((char)c);
        }
        return !out.empty();
    }

    static std::vector<std::string> split(const std::string& s)
    {
        std::vector<std::string> t;
        size_t i = 0;
        while (i < s.size()) {
            while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) i++;
            size_t j = i;
            while (j < s.size() && s[j] != ' ' && s[j] != '\t') j++;
            if (j > i) t.push_back(s.substr(i, j - i));
            i = j;
        }
        return t;
    }

    bool open(const char* path)
    {
        f = fopen(path, "rb");
        if (!f) return fail(std::string("cannot open ") + path);
        std::string line;
        if (!read_line(line) || line != "ply") return fail("not a PLY file (a Git-LFS pointer stub looks like this too)");
        bool haveFormat = false;
        while (read_line(line)) {
            auto t = split(line);
            if (t.empty()) continue;
            if (t[0] == "format") {
                if (t.size() < 2) return fail("bad format line");
                if (t[1] == "ascii") ascii = true;
                else if (t[1] == "binary_little_endian") ascii = false;
                else return fail("unsupported PLY format " + t[1]);
                haveFormat = true;
            } else if (t[0] == "element") {
                if (t.size() < 3) return fail("bad element line");
                Element e; e.name = t[1]; e.count = strtoull(t[2].c_str(), nullptr, 10);
                elements.push_back(e);
            } else if (t[0] == "property") {
                if (elements.empty()) return fail("property before element");
                Prop p;
                if (t.size() >= 5 && t[1] == "list") { p.list = true; p.countType = parse_type(t[2]); p.type = parse_type(t[3]); p.name = t[4]; }
                else if (t.size() >= 3) { p.type = parse_type(t[1]); p.name = t[2]; }
                else return fail("bad property line");
                if (p.type == T_BAD || (p.list && p.countType == T_BAD)) return fail("unknown property type in: " + line);
                elements.back().props.push_back(p);
            } else if (t[0] == "end_header") {
                return haveFormat ? true : fail("no format line");
            }
        }
        return fail("no end_header");
    }

    /* Reads one scalar of type t (binary) or one token (ascii). */
    bool next_number(Type t, double& v)
    {
        if (ascii) {
            char tok[64];
            if (fscanf(f, "%63s", tok) != 1) return false;
            char* end = nullptr;
            v = strtod(tok, &end);
            return end != tok && *end == '\0';                  /* a token that is not a number ends the read */
        }
        unsigned char buf[8];
        size_t n = type_size(t);
        if (fread(buf, 1, n, f) != n) return false;
        v = load_num(buf, t);
        return true;
    }
};

}  // namespace

static int ply_load(const char* path, uint32_t* num_vertices, uint32_t* num_triangles,
                    float* positions, PtVertexAux* aux, uint32_t* indices, char* err, size_t err_len)
{
    auto bail = [&](const std::string& m) { if (err && err_len) snprintf(err, err_len, "pt_ply_load: %s", m.c_str()); return -1; };
    if (!path) return bail("null path");
    Reader r;
    if (!r.open(path)) return bail(r.error);

    uint64_t nv = 0, nt = 0;
    bool haveVertices = false;
    auto to_u8 = [](double d) { return (uint8_t)(d >= 255.0 ? 255.0 : d >= 0.0 ? d : 0.0); };      /* NaN -> 0 */
    for (const Element& e : r.elements) {
        if (e.props.empty() && e.name != "vertex") continue;      /* nothing to read per entry, whatever the count says */
        if (e.name == "vertex") {
            /* a second vertex element would be counted by the sizing call as the only one and written over the first */
            if (haveVertices) return bail("more than one vertex element");
            haveVertices = true;
            if (e.count > 0xFFFFFFFFull) return bail("model too large");
            int ix = -1, iy = -1, iz = -1, inx = -1, iny = -1, inz = -1, ir = -1, ig = -1, ib = -1, ia = -1;
            for (size_t k = 0; k < e.props.size(); k++) {
                const Prop& p = e.props[k];
                if (p.list) return bail("list property in vertex element");
                if (p.name == "x") ix = (int)k; else if (p.name == "y") iy = (int)k; else if (p.name == "z") iz = (int)k;
                else if (p.name == "nx") inx = (int)k; else if (p.name == "ny") iny = (int)k; else if (p.name == "nz") inz = (int)k;
                else if (p.name == "red") ir = (int)k; else if (p.name == "green") ig = (int)k; else if (p.name == "blue") ib = (int)k;
                else if (p.name == "alpha") ia = (int)k;
            }
            if (ix < 0 || iy < 0 || iz < 0) return bail("vertex element lacks x/y/z");
            std::vector<double> row(e.props.size());
            for (uint64_t v = 0; v < e.count; v++) {
                for (size_t k = 0; k < e.props.size(); k++)
                    if (!r.next_number(e.props[k].type, row[k])) return bail("truncated vertex data");
                if (positions) {
                    float* p = positions + 4 * v;
                    p[0] = (float)row[ix]; p[1] = (float)row[iy]; p[2] = (float)row[iz]; p[3] = 1.0f;
                }
                if (aux) {
                    PtVertexAux& a = aux[v];
                    a.normal[0] = inx >= 0 ? (float)row[inx] : 0.0f; a.normal[1] = iny >= 0 ? (float)row[iny] : 0.0f; a.normal[2] = inz >= 0 ? (float)row[inz] : 0.0f;
                    a.color[0] = ir >= 0 ? to_u8(row[ir]) : 255; a.color[1] = ig >= 0 ? to_u8(row[ig]) : 255;
                    a.color[2] = ib >= 0 ? to_u8(row[ib]) : 255; a.color[3] = ia >= 0 ? to_u8(row[ia]) : 255;
                }
            }
            nv = e.count;
        } else if (e.name == "face") {
            std::vector<uint32_t> poly;
            for (uint64_t fc = 0; fc < e.count; fc++) {
                for (const Prop& p : e.props) {
                    if (!p.list) { double d; if (!r.next_number(p.type, d)) return bail("truncated face data"); continue; }
                    double cnt;
                    if (!r.next_number(p.countType, cnt)) return bail("truncated face data");
                    if (!(cnt >= 0.0 && cnt <= 65535.0)) return bail("bad list length in face data");
                    size_t n = (size_t)cnt;
                    poly.resize(n);
                    for (size_t k = 0; k < n; k++) {
                        double d;
                        if (!r.next_number(p.type, d)) return bail("truncated face data");
                        if (!(d >= 0.0 && d <= 4294967295.0)) return bail("face refers to a missing vertex");
                        poly[k] = (uint32_t)d;
                    }
                    if (p.name != "vertex_indices" && p.name != "vertex_index") continue;
                    for (size_t k = 1; k + 1 < n; k++) {
                        if (indices) { indices[3 * nt] = poly[0]; indices[3 * nt + 1] = poly[k]; indices[3 * nt + 2] = poly[k + 1]; }
                        nt++;
                    }
                }
            }
        } else {
            /* skip other elements */
            for (uint64_t i = 0; i < e.count; i++)
                for (const Prop& p : e.props) {
                    double d;
                    if (p.list) {
                        if (!r.next_number(p.countType, d)) return bail("truncated data");
                        if (!(d >= 0.0 && d <= 4294967295.0)) return bail("bad list length");
                        for (size_t k = 0, n = (size_t)d; k < n; k++) if (!r.next_number(p.type, d)) return bail("truncated data");
                    } else if (!r.next_number(p.type, d)) return bail("truncated data");
                }
        }
    }
    if (nv > 0xFFFFFFFFull || nt > 0x2AAAAAAAull) return bail("model too large");
    if (indices)
        for (uint64_t i = 0; i < 3 * nt; i++) if (indices[i] >= nv) return bail("face refers to a missing vertex");
    if (num_vertices) *num_vertices = (uint32_t)nv;
    if (num_triangles) *num_triangles = (uint32_t)nt;
    return 0;
}

extern "C" PT_API int pt_ply_load(const char* path, uint32_t* num_vertices, uint32_t* num_triangles,
                                  float* positions, PtVertexAux* aux, uint32_t* indices, char* err, size_t err_len)
{
    try {
        return ply_load(path, num_vertices, num_triangles, positions, aux, indices, err, err_len);
    } catch (...) {      /* out of memory while reading: nothing may unwind through the C boundary */
        if (err && err_len) snprintf(err, err_len, "pt_ply_load: out of memory");
        return -1;
    }
}
