// Model files: `.moped.xml` parsing and the packed `.mopeddb` container.
//
// Replaces, on the host side of the drop-in boundary,
//   * the sXML tokenizer         moped2/libmoped/include/sXML.hpp:53-118
//   * Moped::addModel(sXML&)     moped2/libmoped/src/moped.cpp:101-137
//     (name = root property "name"; the LAST child called "Points"; one model point per
//      child of it: p3d -> 3 floats, desc -> floats until the stream fails, filed under
//      desc_type; bounding box = min/max of the points)
// with a single pass over the mapped file (no DOM, no std::map per node, no istream),
// and adds what the reference does not have: a binary container that is mapped and
// handed to the GPU as it lies, so a 1M-descriptor database (≈1 GB of decimal text,
// minutes of parsing) loads at storage/PCIe speed.
//
// Number parsing is correctly rounded like the strtof under `istream >> float` (parse_one),
// with the same stop-at-the-first-bad-token behaviour.  Host code only.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/moped_hip.h"

namespace {

constexpr int DIM = MH_DESC_DIM;

struct ModelEntry {
  std::string name;
  uint64_t row_begin = 0, n_rows = 0;
  float bbox[6] = {10E10f, 10E10f, 10E10f, -10E10f, -10E10f, -10E10f};  // moped.cpp:107-108
};

struct Mapping {
  void* base = nullptr;
  size_t bytes = 0;
  ~Mapping() {
    if (base) munmap(base, bytes);
  }
};

}  // namespace

struct mh_model_set {
  std::string desc_type = "SIFT";
  std::vector<ModelEntry> models;
  // owned storage (parsed models) ...
  std::vector<float> desc, xyz;
  std::vector<int32_t> model_of;
  // ... or views into a mapped .mopeddb
  Mapping* map = nullptr;
  const float* desc_p = nullptr;
  const float* xyz_p = nullptr;
  uint64_t n_rows = 0;
  std::string err;
  ~mh_model_set() { delete map; }
  const float* D() const { return map ? desc_p : desc.data(); }
  const float* X() const { return map ? xyz_p : xyz.data(); }
};

namespace {

// ---- XML ------------------------------------------------------------------------------
inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

struct Cursor {
  const char* p;
  const char* end;
  bool eof() const { return p >= end; }
};

// One attribute value: p is just after the opening quote; returns the raw range and leaves
// p after the closing quote.  `escaped` tells whether a backslash occurs (rare: then the
// caller unescapes the way sXML does, sXML.hpp:87-92).
bool attr_value(Cursor& c, const char*& b, const char*& e, bool& escaped) {
  b = c.p;
  escaped = false;
  while (!c.eof() && *c.p != '"') {
    if (*c.p == '\\') {
      escaped = true;
      ++c.p;                      // the character after a backslash is taken literally
      if (c.eof()) return false;
    }
    ++c.p;
  }
  if (c.eof()) return false;
  e = c.p++;
  return true;
}

std::string unescape(const char* b, const char* e) {
  std::string s;
  for (const char* p = b; p < e; ++p) {
    if (*p == '\\' && p + 1 < e) {
      ++p;
      if (*p == 'n') {            // "\n" -> newline, and the NEXT character is appended as is
        s += '\n';
        if (++p >= e) break;
      }
    }
    s += *p;
  }
  return s;
}

// One decimal number -> float, correctly rounded (what the strtof under `istream >> float`
// returns).  Fast path for plain `[-+]ddd.ddd` with at most 19 digits: the digits as an
// integer m and a power of ten 10^k <= 10^22 are both exact doubles, so m / 10^k is the
// correctly rounded double (Clinger); rounding that double to float is exact too unless the
// double sits on a float rounding boundary -- those, exponents, and everything unusual go to
// std::from_chars (libstdc++ 11 runs strtod under a locale switch there: ~10x slower).
inline bool parse_one(const char*& p, const char* e, float& out) {
  static const double P10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const char* q = p;
  bool neg = false;
  if (q < e && (*q == '-' || *q == '+')) {
    neg = *q == '-';
    ++q;
  }
  uint64_t m = 0;
  int nd = 0, frac = 0;
  while (q < e && *q >= '0' && *q <= '9') {
    m = m * 10 + (uint64_t)(*q - '0');
    ++nd;
    ++q;
  }
  if (q < e && *q == '.') {
    ++q;
    while (q < e && *q >= '0' && *q <= '9') {
      m = m * 10 + (uint64_t)(*q - '0');
      ++nd;
      ++frac;
      ++q;
    }
  }
  const bool plain = nd > 0 && nd <= 19 && m < (1ull << 53) && !(q < e && (*q == 'e' || *q == 'E'));
  if (plain) {
    const double d = (double)m / P10[frac];
    uint64_t bits;
    memcpy(&bits, &d, sizeof bits);
    const uint64_t low = bits & ((1ull << 29) - 1);      // bits dropped by the float rounding
    if (low != (1ull << 28) && low != 0) {                // not on (or next to) a tie
      out = neg ? -(float)d : (float)d;
      p = q;
      return true;
    }
    if (low == 0) {                                       // exactly a float (e.g. 0.5, 1.25): exact
      out = neg ? -(float)d : (float)d;
      p = q;
      return true;
    }
  }
  // general path
  const char* g = p;
  if (g < e && *g == '+') ++g;  // istream accepts a leading '+', from_chars does not
  if (g >= e || !((*g >= '0' && *g <= '9') || *g == '.' || *g == '-')) return false;  // no inf/nan, like num_get
  float v;
  const std::from_chars_result r = std::from_chars(g, e, v);
  if (r.ec != std::errc() || r.ptr == g) return false;
  out = v;
  p = r.ptr;
  return true;
}

// Parses floats separated by white space until the first token that is not a number
// (`while( jss >> f )`, moped.cpp:126-128).  Returns how many were stored (at most cap).
int parse_floats(const char* b, const char* e, float* out, int cap, int* total) {
  int n = 0, stored = 0;
  const char* p = b;
  while (true) {
    while (p < e && is_space(*p)) ++p;
    if (p >= e) break;
    float v;
    if (!parse_one(p, e, v)) break;
    if (stored < cap) out[stored++] = v;
    ++n;
  }
  if (total) *total = n;
  return stored;
}

struct Attr {
  const char *nb, *ne, *vb, *ve;
  bool escaped;
};

// Reads `<name attr="v" ...` up to and including `>` or `/>`.  Returns 0 on EOF/garbage,
// 1 = open tag, 2 = self-closed, 3 = closing tag (`</name>`), 4 = comment/declaration skipped.
int read_tag(Cursor& c, const char*& nb, const char*& ne, std::vector<Attr>* attrs) {
  while (!c.eof() && *c.p != '<') ++c.p;
  if (c.eof()) return 0;
  ++c.p;
  if (c.end - c.p >= 3 && c.p[0] == '!' && c.p[1] == '-' && c.p[2] == '-') {  // <!-- ... -->
    c.p += 3;
    while (c.end - c.p >= 3 && !(c.p[0] == '-' && c.p[1] == '-' && c.p[2] == '>')) ++c.p;
    if (c.end - c.p < 3) return 0;
    c.p += 3;
    return 4;
  }
  if (!c.eof() && (*c.p == '?' || *c.p == '!')) {  // <?xml ...?>, <!DOCTYPE ...>
    while (!c.eof() && *c.p != '>') ++c.p;
    if (c.eof()) return 0;
    ++c.p;
    return 4;
  }
  const bool closing = !c.eof() && *c.p == '/';
  nb = c.p;
  while (!c.eof() && !is_space(*c.p) && *c.p != '>' && *c.p != '=' && !(*c.p == '/' && c.p != nb)) ++c.p;
  ne = c.p;
  if (closing) {
    while (!c.eof() && *c.p != '>') ++c.p;
    if (c.eof()) return 0;
    ++c.p;
    return 3;
  }
  if (attrs) attrs->clear();
  while (true) {
    while (!c.eof() && is_space(*c.p)) ++c.p;
    if (c.eof()) return 0;
    if (*c.p == '>') {
      ++c.p;
      return 1;
    }
    if (*c.p == '/') {
      while (!c.eof() && *c.p != '>') ++c.p;
      if (c.eof()) return 0;
      ++c.p;
      return 2;
    }
    Attr a;
    a.nb = c.p;
    while (!c.eof() && !is_space(*c.p) && *c.p != '=' && *c.p != '>' && *c.p != '/') ++c.p;
    a.ne = c.p;
    while (!c.eof() && is_space(*c.p)) ++c.p;
    if (c.eof() || *c.p != '=') continue;  // a bare word: ignored (sXML stops reading properties there)
    while (!c.eof() && *c.p != '"') ++c.p;
    if (c.eof()) return 0;
    ++c.p;
    if (!attr_value(c, a.vb, a.ve, a.escaped)) return 0;
    if (attrs) attrs->push_back(a);
  }
}

inline bool name_is(const char* b, const char* e, const char* s) {
  const size_t n = strlen(s);
  return (size_t)(e - b) == n && memcmp(b, s, n) == 0;
}

// Skips everything up to the end tag matching an element that was just opened.
bool skip_subtree(Cursor& c) {
  int depth = 1;
  const char *nb, *ne;
  while (depth > 0) {
    const int t = read_tag(c, nb, ne, nullptr);
    if (t == 0) return false;
    if (t == 1) ++depth;
    if (t == 3) --depth;
  }
  return true;
}

struct ParsedModel {
  std::string name;
  std::vector<float> xyz, desc;
  float bbox[6];
  int bad_len = 0;
};

bool parse_model_xml(const char* data, size_t bytes, const std::string& desc_type, ParsedModel& out,
                     std::string& err) {
  Cursor c{data, data + bytes};
  std::vector<Attr> attrs;
  const char *nb, *ne;
  int t;
  do t = read_tag(c, nb, ne, &attrs);
  while (t == 4);
  if (t != 1 && t != 2) {
    err = "no root element";
    return false;
  }
  out.name.clear();
  for (const Attr& a : attrs)
    if (name_is(a.nb, a.ne, "name")) out.name = a.escaped ? unescape(a.vb, a.ve) : std::string(a.vb, a.ve);
  const ModelEntry fresh;
  memcpy(out.bbox, fresh.bbox, sizeof out.bbox);
  out.xyz.clear();
  out.desc.clear();
  out.bad_len = 0;
  if (t == 2) return true;  // no children: a model without points (addModel returns "" there)
  // children of the root
  while (true) {
    t = read_tag(c, nb, ne, &attrs);
    if (t == 0 || t == 3) break;
    if (t == 4 || t == 2) continue;
    if (!name_is(nb, ne, "Points")) {
      if (!skip_subtree(c)) break;
      continue;
    }
    // a later <Points> replaces an earlier one (moped.cpp:110-113 keeps the last)
    out.xyz.clear();
    out.desc.clear();
    out.bad_len = 0;
    memcpy(out.bbox, fresh.bbox, sizeof out.bbox);
    while (true) {
      t = read_tag(c, nb, ne, &attrs);
      if (t == 0 || t == 3) break;
      if (t == 4) continue;
      // every child of <Points> is a model point, whatever its tag (moped.cpp:117)
      const Attr *p3d = nullptr, *desc = nullptr, *type = nullptr;
      for (const Attr& a : attrs) {
        if (name_is(a.nb, a.ne, "p3d")) p3d = &a;
        if (name_is(a.nb, a.ne, "desc")) desc = &a;
        if (name_is(a.nb, a.ne, "desc_type")) type = &a;
      }
      float p[3] = {0.f, 0.f, 0.f};
      if (p3d) parse_floats(p3d->vb, p3d->ve, p, 3, nullptr);
      for (int k = 0; k < 3; ++k) {   // the bounding box takes every point, of any descriptor type
        if (p[k] < out.bbox[k]) out.bbox[k] = p[k];
        if (p[k] > out.bbox[3 + k]) out.bbox[3 + k] = p[k];
      }
      const bool mine = type ? name_is(type->vb, type->ve, desc_type.c_str()) : desc_type.empty();
      if (mine) {
        const size_t at = out.desc.size();
        out.desc.resize(at + DIM, 0.f);
        int total = 0;
        if (desc) parse_floats(desc->vb, desc->ve, &out.desc[at], DIM, &total);
        if (total != DIM) ++out.bad_len;
        out.xyz.insert(out.xyz.end(), p, p + 3);
      }
      if (t == 1 && !skip_subtree(c)) return true;  // <Observation> children are not model points
    }
  }
  return true;
}

bool map_file(const char* path, Mapping& m, std::string& err) {
  const int fd = open(path, O_RDONLY);
  if (fd < 0) {
    err = std::string("cannot open ") + path;
    return false;
  }
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size <= 0) {
    close(fd);
    err = std::string("empty or unreadable file ") + path;
    return false;
  }
  void* b = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (b == MAP_FAILED) {
    err = std::string("mmap failed for ") + path;
    return false;
  }
  m.base = b;
  m.bytes = (size_t)st.st_size;
  return true;
}

// ---- .mopeddb ---------------------------------------------------------------------------
// Little endian, sections 4 KiB aligned so they can be mapped and copied to the GPU as is.
//   Header (one 4 KiB page)
//   ModelRecord[n_models]
//   names blob
//   xyz   float[n_rows][3]
//   desc  float[n_rows][128]   raw, as parsed: normalisation happens at upload like Update()
struct DbHeader {
  char magic[8];        // "MOPEDDB1"
  uint32_t version;     // 1
  uint32_t dim;         // 128
  uint32_t n_models;
  uint32_t flags;       // 0
  uint64_t n_rows;
  uint64_t off_models, off_names, names_bytes, off_xyz, off_desc, file_bytes;
  char desc_type[32];
};
struct DbModelRecord {
  uint64_t row_begin, n_rows;
  float bbox[6];
  uint32_t name_off, name_len;
};
constexpr size_t PAGE = 4096;
inline size_t page_up(size_t x) { return (x + PAGE - 1) / PAGE * PAGE; }

}  // namespace

extern "C" {

int mh_models_create(mh_model_set** out, const char* desc_type) {
  if (!out) return MH_ERR_ARG;
  *out = new mh_model_set;
  if (desc_type) (*out)->desc_type = desc_type;
  return MH_OK;
}

void mh_models_destroy(mh_model_set* s) { delete s; }

const char* mh_models_last_error(const mh_model_set* s) { return s ? s->err.c_str() : "null model set"; }

int mh_models_add_xml_buffer(mh_model_set* s, const char* data, int64_t bytes) {
  if (!s || !data || bytes < 0) return MH_ERR_ARG;
  if (s->map) {
    s->err = "model set is a read-only view of a .mopeddb file";
    return MH_ERR_ARG;
  }
  ParsedModel pm;
  if (!parse_model_xml(data, (size_t)bytes, s->desc_type, pm, s->err)) return MH_ERR_ARG;
  if (pm.bad_len) {
    s->err = "model '" + pm.name + "': " + std::to_string(pm.bad_len) + " point(s) whose descriptor does not have " +
             std::to_string(DIM) + " values";
    return MH_ERR_ARG;
  }
  // a model with the name of an existing one replaces it (moped.cpp:141-146); others append
  ModelEntry e;
  e.name = pm.name;
  memcpy(e.bbox, pm.bbox, sizeof e.bbox);
  const uint64_t n = pm.xyz.size() / 3;
  int at = -1;
  for (size_t i = 0; i < s->models.size(); ++i)
    if (s->models[i].name == pm.name) at = (int)i;
  if (at >= 0) {
    // rebuild the flat arrays without the old rows, new rows in the old position
    const ModelEntry old = s->models[at];
    std::vector<float> desc, xyz;
    desc.reserve(s->desc.size() - old.n_rows * DIM + pm.desc.size());
    xyz.reserve(s->xyz.size() - old.n_rows * 3 + pm.xyz.size());
    desc.insert(desc.end(), s->desc.begin(), s->desc.begin() + old.row_begin * DIM);
    desc.insert(desc.end(), pm.desc.begin(), pm.desc.end());
    desc.insert(desc.end(), s->desc.begin() + (old.row_begin + old.n_rows) * DIM, s->desc.end());
    xyz.insert(xyz.end(), s->xyz.begin(), s->xyz.begin() + old.row_begin * 3);
    xyz.insert(xyz.end(), pm.xyz.begin(), pm.xyz.end());
    xyz.insert(xyz.end(), s->xyz.begin() + (old.row_begin + old.n_rows) * 3, s->xyz.end());
    s->desc.swap(desc);
    s->xyz.swap(xyz);
    e.row_begin = old.row_begin;
    e.n_rows = n;
    s->models[at] = e;
    uint64_t run = 0;
    for (ModelEntry& m : s->models) {
      m.row_begin = run;
      run += m.n_rows;
    }
  } else {
    e.row_begin = s->xyz.size() / 3;
    e.n_rows = n;
    s->desc.insert(s->desc.end(), pm.desc.begin(), pm.desc.end());
    s->xyz.insert(s->xyz.end(), pm.xyz.begin(), pm.xyz.end());
    s->models.push_back(e);
  }
  s->n_rows = s->xyz.size() / 3;
  s->model_of.resize(s->n_rows);
  for (size_t i = 0; i < s->models.size(); ++i)
    for (uint64_t r = 0; r < s->models[i].n_rows; ++r) s->model_of[s->models[i].row_begin + r] = (int32_t)i;
  return MH_OK;
}

int mh_models_add_xml(mh_model_set* s, const char* path) {
  if (!s || !path) return MH_ERR_ARG;
  Mapping m;
  if (!map_file(path, m, s->err)) return MH_ERR_ARG;
  return mh_models_add_xml_buffer(s, (const char*)m.base, (int64_t)m.bytes);
}

int mh_models_count(const mh_model_set* s) { return s ? (int)s->models.size() : 0; }
int64_t mh_models_rows(const mh_model_set* s) { return s ? (int64_t)s->n_rows : 0; }
const char* mh_models_name(const mh_model_set* s, int i) {
  return (s && i >= 0 && i < (int)s->models.size()) ? s->models[i].name.c_str() : "";
}
int mh_models_range(const mh_model_set* s, int i, int64_t* row_begin, int64_t* n_rows, float bbox[6]) {
  if (!s || i < 0 || i >= (int)s->models.size()) return MH_ERR_ARG;
  if (row_begin) *row_begin = (int64_t)s->models[i].row_begin;
  if (n_rows) *n_rows = (int64_t)s->models[i].n_rows;
  if (bbox) memcpy(bbox, s->models[i].bbox, sizeof(float) * 6);
  return MH_OK;
}
const float* mh_models_desc(const mh_model_set* s) { return s ? s->D() : nullptr; }
const float* mh_models_xyz(const mh_model_set* s) { return s ? s->X() : nullptr; }

int mh_models_save(const mh_model_set* s, const char* path) {
  if (!s || !path) return MH_ERR_ARG;
  std::string names;
  std::vector<DbModelRecord> rec(s->models.size());
  for (size_t i = 0; i < s->models.size(); ++i) {
    rec[i].row_begin = s->models[i].row_begin;
    rec[i].n_rows = s->models[i].n_rows;
    memcpy(rec[i].bbox, s->models[i].bbox, sizeof rec[i].bbox);
    rec[i].name_off = (uint32_t)names.size();
    rec[i].name_len = (uint32_t)s->models[i].name.size();
    names += s->models[i].name;
    names += '\0';
  }
  DbHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, "MOPEDDB1", 8);
  h.version = 1;
  h.dim = DIM;
  h.n_models = (uint32_t)s->models.size();
  h.n_rows = s->n_rows;
  h.off_models = PAGE;
  h.off_names = h.off_models + rec.size() * sizeof(DbModelRecord);
  h.names_bytes = names.size();
  h.off_xyz = page_up(h.off_names + names.size());
  h.off_desc = page_up(h.off_xyz + s->n_rows * 3 * sizeof(float));
  h.file_bytes = h.off_desc + s->n_rows * DIM * sizeof(float);
  snprintf(h.desc_type, sizeof h.desc_type, "%s", s->desc_type.c_str());
  FILE* f = fopen(path, "wb");
  if (!f) return MH_ERR_ARG;
  bool ok = true;
  const std::vector<char> zeros(PAGE, 0);
  auto pad_to = [&](uint64_t off) {
    const long cur = ftell(f);
    if ((uint64_t)cur < off) ok &= fwrite(zeros.data(), 1, off - cur, f) == off - cur;
  };
  ok &= fwrite(&h, sizeof h, 1, f) == 1;
  pad_to(h.off_models);
  if (!rec.empty()) ok &= fwrite(rec.data(), sizeof(DbModelRecord), rec.size(), f) == rec.size();
  if (!names.empty()) ok &= fwrite(names.data(), 1, names.size(), f) == names.size();
  pad_to(h.off_xyz);
  if (s->n_rows) ok &= fwrite(s->X(), sizeof(float) * 3, s->n_rows, f) == s->n_rows;
  pad_to(h.off_desc);
  if (s->n_rows) ok &= fwrite(s->D(), sizeof(float) * DIM, s->n_rows, f) == s->n_rows;
  ok &= fclose(f) == 0;
  return ok ? MH_OK : MH_ERR_ARG;
}

int mh_models_load(mh_model_set** out, const char* path) {
  if (!out || !path) return MH_ERR_ARG;
  *out = nullptr;
  mh_model_set* s = new mh_model_set;
  s->map = new Mapping;
  if (!map_file(path, *s->map, s->err)) {
    delete s;
    return MH_ERR_ARG;
  }
  const unsigned char* b = (const unsigned char*)s->map->base;
  const size_t bytes = s->map->bytes;
  DbHeader h;
  bool ok = bytes >= sizeof h;
  if (ok) {
    memcpy(&h, b, sizeof h);
    // offsets first, then sizes against what is left behind them: no sum can wrap
    auto fits = [&](uint64_t off, uint64_t count, uint64_t elem) {
      return off <= bytes && count <= (bytes - off) / elem;
    };
    ok = memcmp(h.magic, "MOPEDDB1", 8) == 0 && h.version == 1 && h.dim == (uint32_t)DIM && h.file_bytes <= bytes &&
         h.n_rows < (1ull << 31) && fits(h.off_models, h.n_models, sizeof(DbModelRecord)) &&
         fits(h.off_names, h.names_bytes, 1) && h.off_xyz % 16 == 0 && h.off_desc % 16 == 0 &&
         fits(h.off_xyz, h.n_rows, 12) && fits(h.off_desc, h.n_rows, (uint64_t)DIM * 4);
  }
  if (!ok) {
    delete s;
    return MH_ERR_ARG;
  }
  h.desc_type[sizeof h.desc_type - 1] = 0;
  s->desc_type = h.desc_type;
  s->n_rows = h.n_rows;
  s->xyz_p = (const float*)(b + h.off_xyz);
  s->desc_p = (const float*)(b + h.off_desc);
  s->models.resize(h.n_models);
  s->model_of.resize(h.n_rows);
  uint64_t run = 0;
  for (uint32_t i = 0; i < h.n_models; ++i) {
    DbModelRecord r;
    memcpy(&r, b + h.off_models + (size_t)i * sizeof r, sizeof r);
    if (r.row_begin != run || r.n_rows > h.n_rows - run || r.name_off > h.names_bytes ||
        r.name_len > h.names_bytes - r.name_off) {
      delete s;
      return MH_ERR_ARG;
    }
    s->models[i].name.assign((const char*)b + h.off_names + r.name_off, r.name_len);
    s->models[i].row_begin = r.row_begin;
    s->models[i].n_rows = r.n_rows;
    memcpy(s->models[i].bbox, r.bbox, sizeof r.bbox);
    for (uint64_t k = 0; k < r.n_rows; ++k) s->model_of[r.row_begin + k] = (int32_t)i;
    run += r.n_rows;
  }
  if (run != h.n_rows) {
    delete s;
    return MH_ERR_ARG;
  }
  *out = s;
  return MH_OK;
}

// defined in api.hip
int mh_db_upload_raw(mh_ctx* ctx, const float* desc_host, const int32_t* model_of_host, const float* xyz_host,
                     int N, int n_models, int32_t index_base, int normalize);

int mh_db_upload_models(mh_ctx* ctx, const mh_model_set* s, int first_model, int n_models) {
  if (!ctx || !s || first_model < 0 || n_models < 0 || first_model + n_models > (int)s->models.size())
    return MH_ERR_ARG;
  uint64_t r0 = 0, r1 = 0;
  if (n_models > 0) {
    r0 = s->models[first_model].row_begin;
    r1 = s->models[first_model + n_models - 1].row_begin + s->models[first_model + n_models - 1].n_rows;
  }
  // model ids stay global (rows of another shard are recognised by index_base), so the
  // device table is sized for all models of the set
  return mh_db_upload_raw(ctx, s->D() + r0 * DIM, s->model_of.data() + r0, s->X() + r0 * 3, (int)(r1 - r0),
                          (int)s->models.size(), (int32_t)r0, 1);
}

}  // extern "C"
