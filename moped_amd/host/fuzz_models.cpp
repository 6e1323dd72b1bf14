// Mutation fuzz of the model loaders (csrc/models.cpp) for the sanitizer build (`make asan` in this directory; CPU only,
// no GPU and no HIP runtime: mh_db_upload_raw is stubbed below because nothing here uploads).
//
// The loaders take untrusted input: `.moped.xml` text (they replace the reference's sXML parser,
// moped2/libmoped/include/sXML.hpp:66-118, reached from Moped::addModel, moped2/libmoped/src/moped.cpp:101-137) and the
// packed `.mopeddb` container.  Every iteration mutates a seed file (byte flips, truncation, chunk deletion /
// duplication / splice, digit and quote damage, huge counts), hands it to mh_models_add_xml_buffer or mh_models_load
// and, when the loader accepts it, walks everything the set exposes (names, ranges, every row) and saves / reloads it.
// A finding is a sanitizer report or a failed invariant (exit 1); rejected inputs are the expected outcome.
//
//   fuzz_models <seed.moped.xml> [iterations = 10000] [rng seed = 1]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#include <vector>

#include "moped_hip.h"

extern "C" int mh_db_upload_raw(mh_ctx*, const float*, const int32_t*, const float*, int, int, int32_t, int) {
  return MH_ERR_ARG;   // never reached: the fuzz does not upload
}

namespace {

uint64_t rng_state = 1;
uint64_t rnd() {   // xorshift64*
  rng_state ^= rng_state >> 12;
  rng_state ^= rng_state << 25;
  rng_state ^= rng_state >> 27;
  return rng_state * 0x2545F4914F6CDD1DULL;
}
size_t below(size_t n) { return n ? (size_t)(rnd() % n) : 0; }

std::string read_file(const char* path) {
  std::string s;
  FILE* f = fopen(path, "rb");
  if (!f) return s;
  char buf[65536];
  size_t k;
  while ((k = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, k);
  fclose(f);
  return s;
}

void mutate(std::string& s) {
  const int n_ops = 1 + (int)below(4);
  for (int op = 0; op < n_ops && !s.empty(); ++op) {
    switch (below(10)) {
      case 0:   // flip bytes
        for (int k = 0, n = 1 + (int)below(8); k < n; ++k) s[below(s.size())] ^= (char)(1u << below(8));
        break;
      case 1:   // truncate
        s.resize(below(s.size() + 1));
        break;
      case 2: {   // delete a chunk
        const size_t a = below(s.size()), len = below(std::min<size_t>(s.size() - a, 4096) + 1);
        s.erase(a, len);
        break;
      }
      case 3: {   // duplicate a chunk somewhere else
        const size_t a = below(s.size()), len = below(std::min<size_t>(s.size() - a, 2048) + 1);
        const std::string chunk = s.substr(a, len);
        s.insert(below(s.size() + 1), chunk);
        break;
      }
      case 4: {   // damage the markup
        static const char marks[] = "<>\"/= \n\t&;'";
        s[below(s.size())] = marks[below(sizeof marks - 1)];
        break;
      }
      case 5: {   // replace a number by an extreme one
        static const char* const nums[] = {"1e39", "-1e39", "nan", "inf", "-inf", "1e-46", "0x1p3", "99999999999999999999",
                                           "-", ".", "e", "1e", "+.e+", "4294967296", "-2147483649", ""};
        size_t a = below(s.size());
        while (a < s.size() && !(isdigit((unsigned char)s[a]) || s[a] == '-' || s[a] == '.')) ++a;
        size_t b = a;
        while (b < s.size() && (isdigit((unsigned char)s[b]) || strchr("+-.eE", s[b]))) ++b;
        s.replace(a, b - a, nums[below(sizeof nums / sizeof *nums)]);
        break;
      }
      case 6:   // NUL bytes
        s[below(s.size())] = '\0';
        break;
      case 7: {   // drop or add descriptor values (a point whose descriptor is not 128 long)
        const size_t a = s.find("desc=\"", below(s.size()));
        if (a != std::string::npos) {
          const size_t b = s.find('"', a + 6);
          if (b != std::string::npos && b > a + 8) {
            if (below(2)) s.erase(a + 6, below(b - a - 6));
            else s.insert(a + 6, "0.5 0.25 ");
          }
        }
        break;
      }
      case 8: {   // rename tags / attributes
        static const char* const from[] = {"Model", "Points", "Point", "p3d", "desc_type", "desc", "name", "SIFT"};
        static const char* const to[] = {"Modle", "Point", "Points", "p2d", "desc", "desc_type", "", "SURF"};
        const int k = (int)below(8);
        const size_t a = s.find(from[k], below(s.size()));
        if (a != std::string::npos) s.replace(a, strlen(from[k]), to[k]);
        break;
      }
      default: {   // splice the head of the file onto a random tail
        const size_t a = below(s.size()), b = below(s.size());
        s = s.substr(0, a) + s.substr(b);
        break;
      }
    }
  }
}

int fail(const char* what, long it) {
  fprintf(stderr, "fuzz_models: INVARIANT BROKEN at iteration %ld: %s\n", it, what);
  return 1;
}

// everything a caller can read from an accepted set must be readable and consistent
int walk(const mh_model_set* s, long it) {
  const int n = mh_models_count(s);
  if (n < 0) return fail("negative model count", it);
  const float* D = mh_models_desc(s);
  const float* X = mh_models_xyz(s);
  int64_t expect = 0;
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    const char* name = mh_models_name(s, i);
    if (!name) return fail("model without a name", it);
    sum += (double)strlen(name);
    int64_t b = 0, r = 0;
    float bbox[6];
    if (mh_models_range(s, i, &b, &r, bbox) != MH_OK) return fail("mh_models_range refused a valid index", it);
    if (b != expect || r < 0) return fail("model row ranges are not contiguous", it);
    expect += r;
    for (int64_t k = b; k < b + r; ++k) {
      for (int d = 0; d < MH_DESC_DIM; ++d) sum += D[k * MH_DESC_DIM + d];
      sum += X[3 * k] + X[3 * k + 1] + X[3 * k + 2];
    }
  }
  if (*mh_models_name(s, n) || *mh_models_name(s, -1) || mh_models_range(s, n, nullptr, nullptr, nullptr) == MH_OK)
    return fail("a name or a range for an index outside the set", it);
  if (mh_models_rows(s) != expect) return fail("mh_models_rows disagrees with the model table", it);
  volatile double sink = sum;
  (void)sink;
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: fuzz_models <seed.moped.xml> [iterations] [rng seed]\n");
    return 2;
  }
  const std::string seed_xml = read_file(argv[1]);
  if (seed_xml.empty()) {
    fprintf(stderr, "fuzz_models: cannot read %s\n", argv[1]);
    return 2;
  }
  const long iters = argc > 2 ? atol(argv[2]) : 10000;
  rng_state = argc > 3 ? strtoull(argv[3], nullptr, 10) * 0x9E3779B97F4A7C15ULL + 1 : 1;
  char tmpl[] = "/tmp/fuzz_models_XXXXXX";
  const char* dir = mkdtemp(tmpl);
  if (!dir) return 2;
  const std::string db_path = std::string(dir) + "/seed.mopeddb", mut_path = std::string(dir) + "/mut.mopeddb",
                    resave_path = std::string(dir) + "/resave.mopeddb";

  // The seed (or, for tests/golden/models/quirks.moped.xml whose one point with a short descriptor is refused on purpose,
  // the seed without its "oops" line) must load; its container is the second seed.  The XML mutants start from the
  // file as it is, refused point included.
  mh_model_set* s0 = nullptr;
  if (mh_models_create(&s0, "SIFT") != MH_OK) return 2;
  std::string good_xml = seed_xml;
  if (mh_models_add_xml_buffer(s0, good_xml.data(), (int64_t)good_xml.size()) != MH_OK) {
    good_xml.clear();
    for (size_t a = 0; a < seed_xml.size();) {
      size_t b = seed_xml.find('\n', a);
      b = b == std::string::npos ? seed_xml.size() : b + 1;
      if (seed_xml.substr(a, b - a).find("oops") == std::string::npos)
        good_xml.append(seed_xml, a, b - a);
      a = b;
    }
    if (mh_models_add_xml_buffer(s0, good_xml.data(), (int64_t)good_xml.size()) != MH_OK) {
      fprintf(stderr, "fuzz_models: the seed itself is rejected: %s\n", mh_models_last_error(s0));
      mh_models_destroy(s0);
      return 2;
    }
  }
  if (walk(s0, -1)) return 1;
  if (mh_models_save(s0, db_path.c_str()) != MH_OK) return fail("the seed set does not save", -1);
  const std::string seed_db = read_file(db_path.c_str());
  const int seed_models = mh_models_count(s0);
  mh_models_destroy(s0);

  long accepted_xml = 0, accepted_db = 0, n_xml = 0, n_db = 0;
  for (long it = 0; it < iters; ++it) {
    if (it % 3 != 2) {   // XML text
      ++n_xml;
      std::string m = seed_xml;
      mutate(m);
      mh_model_set* s = nullptr;
      const bool sift = below(8) != 0;   // (a set of another descriptor type reads the other points of the file)
      if (mh_models_create(&s, sift ? "SIFT" : "SURF") != MH_OK) return 2;
      // an exact-size heap copy without a terminator: the parser must not read past `bytes`
      char* buf = (char*)malloc(m.size() ? m.size() : 1);
      memcpy(buf, m.data(), m.size());
      const int rc = mh_models_add_xml_buffer(s, buf, (int64_t)m.size());
      free(buf);
      if (rc == MH_OK) {
        ++accepted_xml;
        if (walk(s, it)) return 1;
        // a second model (the seed) beside or instead of it, then save / load / compare
        if (mh_models_add_xml_buffer(s, good_xml.data(), (int64_t)good_xml.size()) != MH_OK && sift) {
          fprintf(stderr, "fuzz_models: %s\n", mh_models_last_error(s));
          return fail("the seed is rejected after an accepted mutant", it);
        }
        if (walk(s, it)) return 1;
        if (mh_models_save(s, resave_path.c_str()) != MH_OK) return fail("an accepted set does not save", it);
        mh_model_set* r = nullptr;
        if (mh_models_load(&r, resave_path.c_str()) != MH_OK) return fail("a saved set does not load", it);
        if (mh_models_count(r) != mh_models_count(s)) return fail("model count changed across save / load", it);
        if (walk(r, it)) return 1;
        const int n = mh_models_count(s);
        int64_t b = 0, rows = 0, b2 = 0, rows2 = 0;
        float bb[6], bb2[6];
        for (int i = 0; i < n; ++i) {
          mh_models_range(s, i, &b, &rows, bb);
          mh_models_range(r, i, &b2, &rows2, bb2);
          if (b != b2 || rows != rows2 || strcmp(mh_models_name(s, i), mh_models_name(r, i)))
            return fail("model table changed across save / load", it);
        }
        if (n && b + rows > 0 && memcmp(mh_models_desc(s), mh_models_desc(r), (size_t)(b + rows) * MH_DESC_DIM * sizeof(float)))
          return fail("descriptor rows changed across save / load", it);
        mh_models_destroy(r);
      } else if (!mh_models_last_error(s) || !*mh_models_last_error(s)) {
        return fail("a rejected input left no error text", it);
      }
      mh_models_destroy(s);
    } else {   // the packed container
      ++n_db;
      std::string m = seed_db;
      // (mostly header-sized damage: a flipped payload float is still a valid file)
      if (below(3)) {
        const size_t head = std::min<size_t>(m.size(), 8192);
        for (int k = 0, n = 1 + (int)below(6); k < n; ++k) m[below(head)] ^= (char)(1u << below(8));
        if (!below(4)) m.resize(below(m.size() + 1));
      } else {
        mutate(m);
      }
      FILE* f = fopen(mut_path.c_str(), "wb");
      if (!f) return 2;
      fwrite(m.data(), 1, m.size(), f);
      fclose(f);
      mh_model_set* r = nullptr;
      if (mh_models_load(&r, mut_path.c_str()) == MH_OK) {
        ++accepted_db;
        if (!r) return fail("mh_models_load returned OK and no set", it);
        if (walk(r, it)) return 1;
        // a read-only view refuses additions
        if (mh_models_add_xml_buffer(r, good_xml.data(), (int64_t)good_xml.size()) == MH_OK)
          return fail("a mapped .mopeddb view accepted a new model", it);
        mh_models_destroy(r);
      } else if (r) {
        return fail("mh_models_load failed and still returned a set", it);
      }
    }
  }
  unlink(db_path.c_str());
  unlink(mut_path.c_str());
  unlink(resave_path.c_str());
  rmdir(dir);
  printf("fuzz_models: %ld iterations (%ld xml: %ld accepted; %ld mopeddb: %ld accepted), seed set %d model(s): no finding\n",
         iters, n_xml, accepted_xml, n_db, accepted_db, seed_models);
  return 0;
}
