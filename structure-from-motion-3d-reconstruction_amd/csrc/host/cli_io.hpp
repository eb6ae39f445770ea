// cli_io.hpp — the reference CLI's on-disk surface: config.json (cpp/include/minijson.hpp),
// Middlebury *_par.txt / *_ang.txt (T:111-152) and binary PGM (cpp/include/pgm_io.hpp:24-54).
// Error strings follow the reference so that `ERROR: ...` lines match.
#pragma once
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <fstream>
#include <map>
#include <memory>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "host_math.hpp"

namespace sfmx_cli {

// ------------------------------------------------------------------------------------------ JSON
// config.json reader (the reference reads it with cpp/include/minijson.hpp; behaviour kept: first occurrence of a
// duplicate key wins, numbers go through strtod, \uXXXX becomes UTF-8 without surrogate pairing, and the error texts,
// which reach the user behind "Failed to parse config.json: <path> | ").  Structure: a single-pass scanner that
// fills a flat node table -- containers are kept on an explicit stack, nothing recurses.
class Json {
 public:
  enum class Kind : std::uint8_t { Null, Bool, Number, String, Object, Array };
  struct Node {
    Kind kind = Kind::Null;
    bool flag = false;
    double number = 0.0;
    std::string text;             // String: the value
    std::string key;              // member name when the parent is an Object
    int parent = -1, next = -1;   // next sibling in source order
    int first = -1, last = -1;    // children of Object / Array
  };
  // view of one value inside a parsed document
  struct Ref {
    const Json* doc = nullptr;
    int id = -1;
    explicit operator bool() const { return doc != nullptr && id >= 0; }
    const Node& node() const { return doc->nodes_[(size_t)id]; }
    Ref member(const char* name) const {  // first member with that name (later duplicates were never stored)
      if (!*this || node().kind != Kind::Object) return {};
      for (int c = node().first; c >= 0; c = doc->nodes_[(size_t)c].next)
        if (doc->nodes_[(size_t)c].key == name) return {doc, c};
      return {};
    }
  };
  Ref root() const { return {this, nodes_.empty() ? -1 : 0}; }

  static Json parse(const std::string& text) {
    Json doc;
    Scanner sc{text.data(), text.data() + text.size()};
    std::vector<int> open;  // containers being filled, innermost last
    std::string pending_key;
    bool have_key = false;
    // state: what may come next inside the innermost container
    enum class Expect { Value, FirstMemberOrEnd, Member, FirstElementOrEnd, CommaOrEnd } expect = Expect::Value;
    auto attach = [&](Node&& n) -> int {
      const int id = (int)doc.nodes_.size();
      bool keep = true;
      if (!open.empty()) {
        n.parent = open.back();
        Node& par = doc.nodes_[(size_t)n.parent];
        if (par.kind == Kind::Object) {
          n.key = std::move(pending_key);
          for (int c = par.first; c >= 0; c = doc.nodes_[(size_t)c].next)
            if (doc.nodes_[(size_t)c].key == n.key) keep = false;  // duplicate key: the first one stays
        }
      }
      doc.nodes_.push_back(std::move(n));
      if (keep && !open.empty()) {
        Node& par = doc.nodes_[(size_t)open.back()];
        if (par.last >= 0) doc.nodes_[(size_t)par.last].next = id;
        else par.first = id;
        par.last = id;
      }
      have_key = false;
      return id;
    };
    for (;;) {
      sc.skip_space();
      if (expect == Expect::Value || expect == Expect::FirstElementOrEnd) {
        if (expect == Expect::FirstElementOrEnd && sc.take(']')) {
          open.pop_back();
        } else {
          if (sc.at_end()) sc.fail("Unexpected end of input");
          const char c = sc.peek();
          Node n;
          if (c == '{' || c == '[') {
            sc.take(c);
            n.kind = c == '{' ? Kind::Object : Kind::Array;
            open.push_back(attach(std::move(n)));
            expect = c == '{' ? Expect::FirstMemberOrEnd : Expect::FirstElementOrEnd;
            continue;
          }
          if (c == '"') { n.kind = Kind::String; n.text = sc.quoted(); }
          else if (c == 'n') { if (!sc.literal("null")) sc.fail("Invalid token (expected null)"); }
          else if (c == 't' || c == 'f') {
            n.kind = Kind::Bool;
            if (sc.literal("true")) n.flag = true;
            else if (!sc.literal("false")) sc.fail("Invalid token (expected true/false)");
          }
          else if (c == '-' || (c >= '0' && c <= '9')) { n.kind = Kind::Number; n.number = sc.number(); }
          else sc.fail(std::string("Unexpected character '") + c + "'");
          attach(std::move(n));
        }
      } else if (expect == Expect::FirstMemberOrEnd || expect == Expect::Member) {
        if (expect == Expect::FirstMemberOrEnd && sc.take('}')) {
          open.pop_back();
        } else {
          if (sc.at_end() || sc.peek() != '"') sc.fail("Expected string key");
          pending_key = sc.quoted();
          have_key = true;
          sc.skip_space();
          if (!sc.take(':')) sc.fail("Expected ':'");
          expect = Expect::Value;
          continue;
        }
      } else {  // CommaOrEnd
        const bool in_object = doc.nodes_[(size_t)open.back()].kind == Kind::Object;
        if (sc.take(in_object ? '}' : ']')) {
          open.pop_back();
        } else {
          if (!sc.take(',')) sc.fail("Expected ','");
          expect = in_object ? Expect::Member : Expect::Value;
          continue;
        }
      }
      // a complete value has just ended
      if (open.empty()) break;
      expect = Expect::CommaOrEnd;
    }
    (void)have_key;
    sc.skip_space();
    if (!sc.at_end()) throw std::runtime_error("Trailing characters after JSON");
    return doc;
  }

 private:
  struct Scanner {
    const char* cur;
    const char* end;
    [[noreturn]] void fail(const std::string& what) const { throw std::runtime_error("JSON parse error: " + what); }
    bool at_end() const { return cur >= end; }
    char peek() const { return *cur; }
    void skip_space() { while (cur < end && std::isspace((unsigned char)*cur)) ++cur; }
    bool take(char c) {
      if (cur < end && *cur == c) { ++cur; return true; }
      return false;
    }
    bool literal(const char* word) {
      const size_t n = std::strlen(word);
      if ((size_t)(end - cur) < n || std::memcmp(cur, word, n) != 0) return false;
      cur += n;
      return true;
    }
    static void append_utf8(std::string& out, unsigned cp) {  // code points below 0x10000 only (no surrogate pairing)
      if (cp < 0x80) { out += (char)cp; return; }
      if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); return; }
      out += (char)(0xE0 | (cp >> 12));
      out += (char)(0x80 | ((cp >> 6) & 0x3F));
      out += (char)(0x80 | (cp & 0x3F));
    }
    std::string quoted() {
      if (!take('"')) fail("Expected '\"'");
      static const char kEsc[] = "\"\\/bfnrt", kRaw[] = "\"\\/\b\f\n\r\t";
      std::string out;
      while (cur < end) {
        const char c = *cur++;
        if (c == '"') return out;
        if (c != '\\') { out += c; continue; }
        if (cur >= end) fail("Bad escape");
        const char e = *cur++;
        if (e == 'u') {
          if (end - cur < 4) fail("Bad \\u escape");
          unsigned cp = 0;
          for (int i = 0; i < 4; i++) {
            const char h = cur[i];
            const int d = (h >= '0' && h <= '9') ? h - '0' : (h >= 'a' && h <= 'f') ? h - 'a' + 10 : (h >= 'A' && h <= 'F') ? h - 'A' + 10 : -1;
            if (d < 0) fail("Bad hex in \\u escape");
            cp = cp * 16 + (unsigned)d;
          }
          cur += 4;
          append_utf8(out, cp);
          continue;
        }
        const char* hit = e ? std::strchr(kEsc, e) : nullptr;
        if (!hit) fail("Unknown escape");
        out += kRaw[hit - kEsc];
      }
      fail("Unterminated string");
    }
    // JSON number grammar, then strtod on exactly the matched characters
    double number() {
      const char* start = cur;
      auto digits = [&]() { const char* s0 = cur; while (cur < end && *cur >= '0' && *cur <= '9') ++cur; return cur != s0; };
      (void)take('-');
      if (cur >= end) fail("Bad number");
      if (*cur == '0') ++cur;
      else if (!digits()) fail("Bad number");
      if (take('.') && !digits()) fail("Bad fraction");
      if (cur < end && (*cur == 'e' || *cur == 'E')) {
        ++cur;
        if (cur < end && (*cur == '+' || *cur == '-')) ++cur;
        if (!digits()) fail("Bad exponent");
      }
      const std::string lexeme(start, cur);
      char* stop = nullptr;
      const double v = std::strtod(lexeme.c_str(), &stop);
      if (stop == lexeme.c_str()) fail("Bad number conversion");
      return v;
    }
  };
  std::vector<Node> nodes_;
};

// config lookup of the reference (T:65-106): cpp.<section>.<key> overrides common.<section>.<key>
inline Json::Ref config_value(const Json& doc, const char* section, const char* key) {
  for (const char* top : {"cpp", "common"})
    if (Json::Ref v = doc.root().member(top).member(section).member(key)) return v;
  return {};
}
inline std::optional<double> as_number(Json::Ref v) {
  if (v && v.node().kind == Json::Kind::Number) return v.node().number;
  return std::nullopt;
}
inline std::optional<int> as_int(Json::Ref v) {  // T:84-88: llround
  if (const auto d = as_number(v)) return (int)std::llround(*d);
  return std::nullopt;
}
inline std::optional<std::string> as_string(Json::Ref v) {
  if (v && v.node().kind == Json::Kind::String) return v.node().text;
  return std::nullopt;
}

// ------------------------------------------------------------------------------------------ dataset files
struct MBRecord {
  std::string img;
  sfmx_host::Mat3 K, Rwc;
  sfmx_host::V3 twc;
};
// T:120-140
inline std::vector<MBRecord> read_par(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("Failed to open: " + path);
  int n = 0;
  f >> n;
  std::vector<MBRecord> recs;
  recs.reserve((size_t)std::max(0, n));
  for (int i = 0; i < n; i++) {
    MBRecord r;
    f >> r.img;
    double v[21] = {0};
    for (double& x : v) f >> x;
    for (int k = 0; k < 9; k++) { r.K.a[k] = v[k]; r.Rwc.a[k] = v[9 + k]; }
    r.twc = {v[18], v[19], v[20]};
    recs.push_back(r);
  }
  return recs;
}
struct MBAngle { double lat = 0, lon = 0; };
// T:142-152 (first occurrence of a name wins, as with unordered_map::emplace)
inline std::unordered_map<std::string, MBAngle> read_ang(const std::string& path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("Failed to open: " + path);
  std::unordered_map<std::string, MBAngle> a;
  std::string img;
  double lat, lon;
  while (f >> lat >> lon >> img) a.emplace(img, MBAngle{lat, lon});
  return a;
}

struct Gray {
  int w = 0, h = 0;
  std::vector<std::uint8_t> pix;
};

// Binary PGM as the reference accepts it (cpp/include/pgm_io.hpp:24-54): "P5", width, height, maxval 255, ONE separator
// byte, w*h pixels.  The reference reads the header with formatted stream extraction, and its quirks are part of the
// surface: a '#' comment is only recognised where it starts directly behind the previous token (no newline in between);
// a header number that does not parse leaves everything after it at 0, which surfaces as the maxval error.  Here the
// file is read whole and scanned by hand with those rules.
inline Gray read_pgm(const std::string& path) {
  std::string bytes;
  {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Failed to open: " + path);
    bytes.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  }
  struct Header {
    const std::string& s;
    size_t at = 0;
    bool failed = false;  // formatted extraction failed once: every later read yields 0 / nothing
    void blanks() { while (at < s.size() && std::isspace((unsigned char)s[at])) ++at; }
    std::string word() {
      blanks();
      const size_t b0 = at;
      while (at < s.size() && !std::isspace((unsigned char)s[at])) ++at;
      if (at == b0) failed = true;
      return s.substr(b0, at - b0);
    }
    void comments() {  // only a '#' that is the very next byte counts
      while (!failed && at < s.size() && s[at] == '#') {
        while (at < s.size() && s[at] != '\n') ++at;
        if (at < s.size()) ++at;
      }
    }
    int integer() {
      if (failed) return 0;
      blanks();
      size_t p = at;
      bool neg = false;
      if (p < s.size() && (s[p] == '+' || s[p] == '-')) neg = s[p++] == '-';
      long long v = 0;
      const size_t d0 = p;
      while (p < s.size() && s[p] >= '0' && s[p] <= '9' && v < (1ll << 40)) v = v * 10 + (s[p++] - '0');
      if (p == d0 || v > 2147483647ll) { failed = true; return 0; }
      at = p;
      return (int)(neg ? -v : v);
    }
  } hd{bytes};
  if (hd.word() != "P5") throw std::runtime_error("Only binary PGM (P5) supported: " + path);
  hd.comments();
  Gray im;
  im.w = hd.integer();
  hd.comments();
  im.h = hd.integer();
  hd.comments();
  if (hd.integer() != 255) throw std::runtime_error("Only 8-bit PGM supported: " + path);
  const size_t body = hd.at + 1;  // exactly one separator byte
  const size_t want = (im.w > 0 && im.h > 0) ? (size_t)im.w * (size_t)im.h : 0;
  if (im.w < 0 || im.h < 0 || body > bytes.size() || bytes.size() - body < want) throw std::runtime_error("PGM read failed: " + path);
  im.pix.assign(bytes.begin() + (std::ptrdiff_t)body, bytes.begin() + (std::ptrdiff_t)(body + want));
  return im;
}

}  // namespace sfmx_cli
