// yaml_loader.cpp — the reference's scene DSL (yaml.rs:22-509) on top of a small
// YAML-subset reader (the reference uses yaml-rust 0.4.5, Cargo.lock:729; nothing of
// it is available here).  Supported YAML: block sequences and mappings, compact
// "- key: value" items, flow sequences/mappings, plain / 'single' / "double" quoted
// scalars, '#' comments.  Scalars are typed with yaml-rust's rules (Yaml::from_str):
// Integer (i64, 0x.., 0o.., +N), Real (Rust f64 syntax, .inf/.nan), Null, Boolean,
// else String; quoted scalars are always String.
//
// Where the reference panics (unwrap() on a missing key / wrong type, yaml.rs:25,
// 142,187,210,213,331) this loader returns an error string instead.
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <map>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

#include "rt_world.h"

namespace rth {
namespace {

// ------------------------------------------------------------------ YAML subset
// The tree is built over the caller's text: lines and plain scalars are views into it, nodes come from a block arena and
// only quoted scalars with escapes own storage (a 2 MB scene is 75 000 lines and 180 000 nodes; one heap string and one
// reference-counted node each made the reader the slowest step between a scene file and its first pixel).
typedef std::string_view SV;
struct Y;
typedef Y* YP;
struct Y {
    enum T { Bad, Scalar, Seq, Map } t = Bad;
    SV s; bool quoted = false;                           // Scalar
    mutable signed char ty = -1;                         // cached classify() result with its value
    mutable long long iv = 0; mutable double dv = 0.0;
    std::vector<YP> items;                               // Seq
    std::vector<std::pair<YP, YP>> kv;                   // Map, insertion order (LinkedHashMap)
};
struct ParseError { std::string msg; };

struct Line { int indent; SV text; int no; };

SV rtrim(SV s) {
    size_t e = s.size();
    while (e > 0 && (s[e - 1] == ' ' || s[e - 1] == '\t' || s[e - 1] == '\r')) e--;
    return s.substr(0, e);
}
SV ltrim(SV s) {
    size_t b = 0;
    while (b < s.size() && (s[b] == ' ' || s[b] == '\t')) b++;
    return s.substr(b);
}
// strip a trailing comment that is outside quotes
SV strip_comment(SV s) {
    if (s.find('#') == SV::npos) return s;
    char q = 0;
    for (size_t i = 0; i < s.size(); i++) {
        char c = s[i];
        if (q) {
            if (q == '"' && c == '\\') { i++; continue; }
            if (c == q) {
                if (q == '\'' && i + 1 < s.size() && s[i + 1] == '\'') { i++; continue; }
                q = 0;
            }
        } else if (c == '"' || c == '\'') {
            // a quote only opens a quoted scalar at the start of a token
            size_t j = i; while (j > 0 && s[j - 1] == ' ') j--;
            if (j == 0 || i != j || s[j - 1] == ':' || s[j - 1] == '-' || s[j - 1] == ',' || s[j - 1] == '[' || s[j - 1] == '{') q = c;
        } else if (c == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) {
            return s.substr(0, i);
        }
    }
    return s;
}
bool is_dash(SV t) { return t == "-" || t.compare(0, 2, "- ") == 0; }

struct Reader {
    std::vector<Line> lines;
    size_t pos = 0;
    static constexpr size_t kBlock = 4096;
    std::vector<std::unique_ptr<Y[]>> blocks;            // node arena: freed with the reader
    size_t used = kBlock;
    std::deque<std::string> owned;                       // unescaped quoted scalars (stable addresses)

    Y* node(Y::T t) {
        if (used == kBlock) { blocks.emplace_back(new Y[kBlock]); used = 0; }
        Y* y = &blocks.back()[used++];
        y->t = t;
        return y;
    }

    explicit Reader(SV text) {
        size_t n_lines = 1;
        for (char c : text) n_lines += c == '\n';
        lines.reserve(n_lines);
        size_t i = 0; int no = 0;
        while (i <= text.size()) {
            size_t e = text.find('\n', i);
            if (e == SV::npos) e = text.size();
            SV raw = text.substr(i, e - i);
            i = e + 1; no++;
            SV t = rtrim(strip_comment(raw));
            size_t ind = 0; while (ind < t.size() && t[ind] == ' ') ind++;
            if (ind == t.size()) continue;               // blank / comment-only
            if (t.compare(ind, 3, "---") == 0 && ind == 0) { if (!lines.empty()) break; continue; }   // first document only
            if (t.compare(0, 3, "...") == 0) break;
            lines.push_back({(int)ind, t.substr(ind), no});
        }
    }
    [[noreturn]] void err(const std::string& m, int no) { throw ParseError{"yaml line " + std::to_string(no) + ": " + m}; }

    // position of the mapping colon (": " or ':' at end) outside quotes/brackets, or npos
    static size_t find_colon(SV s) {
        char q = 0; int depth = 0;
        for (size_t i = 0; i < s.size(); i++) {
            char c = s[i];
            if (q) {
                if (q == '"' && c == '\\') { i++; continue; }
                if (c == q) q = 0;
                continue;
            }
            if ((c == '"' || c == '\'') && (i == 0)) { q = c; continue; }
            if (c == '[' || c == '{') depth++;
            else if (c == ']' || c == '}') depth--;
            else if (c == ':' && depth == 0 && (i + 1 == s.size() || s[i + 1] == ' ')) return i;
        }
        return SV::npos;
    }

    YP scalar(SV raw) {
        YP y = node(Y::Scalar);
        SV s = rtrim(ltrim(raw));
        if (s.size() >= 2 && s[0] == '"' && s.back() == '"') {
            y->quoted = true;
            if (s.find('\\') == SV::npos) { y->s = s.substr(1, s.size() - 2); return y; }
            std::string o;
            for (size_t i = 1; i + 1 < s.size(); i++) {
                if (s[i] == '\\' && i + 2 < s.size()) {
                    char c = s[++i];
                    o += c == 'n' ? '\n' : c == 't' ? '\t' : c;
                } else o += s[i];
            }
            owned.push_back(std::move(o)); y->s = owned.back();
        } else if (s.size() >= 2 && s[0] == '\'' && s.back() == '\'') {
            y->quoted = true;
            if (s.find('\'', 1) == s.size() - 1) { y->s = s.substr(1, s.size() - 2); return y; }
            std::string o;
            for (size_t i = 1; i + 1 < s.size(); i++) {
                if (s[i] == '\'' && s[i + 1] == '\'' && i + 2 < s.size()) i++;
                o += s[i];
            }
            owned.push_back(std::move(o)); y->s = owned.back();
        } else {
            y->s = s;
        }
        return y;
    }

    // flow value starting at s[i]
    YP flow(SV s, size_t& i, int no) {
        while (i < s.size() && s[i] == ' ') i++;
        if (i >= s.size()) err("unexpected end of flow collection", no);
        if (s[i] == '[') {
            YP y = node(Y::Seq); i++;
            for (;;) {
                while (i < s.size() && s[i] == ' ') i++;
                if (i >= s.size()) err("unterminated flow sequence", no);
                if (s[i] == ']') { i++; break; }
                y->items.push_back(flow(s, i, no));
                while (i < s.size() && s[i] == ' ') i++;
                if (i < s.size() && s[i] == ',') i++;
                else if (i < s.size() && s[i] == ']') { i++; break; }
                else err("expected ',' or ']' in flow sequence", no);
            }
            return y;
        }
        if (s[i] == '{') {
            YP y = node(Y::Map); i++;
            for (;;) {
                while (i < s.size() && s[i] == ' ') i++;
                if (i >= s.size()) err("unterminated flow mapping", no);
                if (s[i] == '}') { i++; break; }
                size_t b = i; char q = 0;
                while (i < s.size() && (q || (s[i] != ':' && s[i] != ',' && s[i] != '}'))) {
                    if (!q && (s[i] == '"' || s[i] == '\'') && i == b) q = s[i]; else if (q && s[i] == q) q = 0;
                    i++;
                }
                YP k = scalar(s.substr(b, i - b));
                YP v;
                if (i < s.size() && s[i] == ':') { i++; v = flow(s, i, no); }
                else v = null_scalar();
                put(y, k, v);
                while (i < s.size() && s[i] == ' ') i++;
                if (i < s.size() && s[i] == ',') i++;
                else if (i < s.size() && s[i] == '}') { i++; break; }
                else err("expected ',' or '}' in flow mapping", no);
            }
            return y;
        }
        size_t b = i; char q = 0;
        if (s[i] == '"' || s[i] == '\'') {
            q = s[i]; i++;
            for (; i < s.size(); i++) {
                if (q == '"' && s[i] == '\\') i++;
                else if (s[i] == q) { if (q == '\'' && i + 1 < s.size() && s[i + 1] == '\'') i++; else break; }
            }
            if (i < s.size()) i++;
            if (i > s.size()) i = s.size();              // a backslash as the last character
            return scalar(s.substr(b, i - b));
        }
        while (i < s.size() && s[i] != ',' && s[i] != ']' && s[i] != '}') i++;
        return scalar(s.substr(b, i - b));
    }

    static void put(YP m, YP k, YP v) {
        // LinkedHashMap::insert on an existing key replaces it and moves it to the back
        for (size_t i = 0; i < m->kv.size(); i++)
            if (m->kv[i].first->t == Y::Scalar && k->t == Y::Scalar && m->kv[i].first->s == k->s &&
                m->kv[i].first->quoted == k->quoted) { m->kv.erase(m->kv.begin() + i); break; }
        m->kv.push_back({k, v});
    }

    YP inline_value(SV text, int no) {
        SV s = ltrim(text);
        if (!s.empty() && (s[0] == '[' || s[0] == '{')) {
            size_t i = 0; YP y = flow(s, i, no);
            if (!rtrim(s.substr(i)).empty()) err("trailing characters after flow collection", no);
            return y;
        }
        return scalar(s);
    }

    YP null_scalar() { YP y = node(Y::Scalar); y->s = "~"; return y; }

    // parse the block whose first line is lines[pos] and whose indentation is `indent`
    YP block(int indent) {
        if (pos >= lines.size()) return null_scalar();
        Line& L = lines[pos];
        if (is_dash(L.text)) return sequence(indent);
        if (find_colon(L.text) != SV::npos) return mapping(indent);
        YP y = inline_value(L.text, L.no); pos++;
        return y;
    }
    YP sequence(int indent) {
        YP y = node(Y::Seq);
        while (pos < lines.size() && lines[pos].indent == indent && is_dash(lines[pos].text)) {
            Line& L = lines[pos];
            SV rest = L.text.substr(1);
            size_t sp = 0; while (sp < rest.size() && rest[sp] == ' ') sp++;
            if (sp == rest.size()) {
                pos++;
                if (pos < lines.size() && lines[pos].indent > indent) y->items.push_back(block(lines[pos].indent));
                else y->items.push_back(null_scalar());
            } else {
                // re-interpret the remainder as a line of its own, indented past the dash
                L.indent = indent + 1 + (int)sp; L.text = rest.substr(sp);
                y->items.push_back(block(L.indent));
            }
        }
        if (pos < lines.size() && lines[pos].indent > indent) err("bad indentation in sequence", lines[pos].no);
        return y;
    }
    YP mapping(int indent) {
        YP y = node(Y::Map);
        while (pos < lines.size() && lines[pos].indent == indent) {
            Line& L = lines[pos];
            if (is_dash(L.text)) break;
            size_t c = find_colon(L.text);
            if (c == SV::npos) err("expected 'key: value'", L.no);
            YP k = scalar(L.text.substr(0, c));
            SV rest = ltrim(L.text.substr(c + 1));
            YP v;
            if (!rest.empty()) { v = inline_value(rest, L.no); pos++; }
            else {
                pos++;
                if (pos < lines.size() && lines[pos].indent > indent) v = block(lines[pos].indent);
                else if (pos < lines.size() && lines[pos].indent == indent && is_dash(lines[pos].text)) v = sequence(indent);
                else v = null_scalar();
            }
            put(y, k, v);
        }
        if (pos < lines.size() && lines[pos].indent > indent) err("bad indentation in mapping", lines[pos].no);
        return y;
    }
    YP document() {
        if (lines.empty()) return null_scalar();
        YP y = block(lines[0].indent);
        if (pos < lines.size()) err("unexpected content", lines[pos].no);
        return y;
    }
};

// ------------------------------------------------------------------ yaml-rust scalar typing
bool parse_i64(SV s, int base, long long& out) {
    if (s.empty()) return false;
    size_t i = 0; bool negv = false;
    if (s[0] == '+' || s[0] == '-') { negv = s[0] == '-'; i = 1; }
    if (i >= s.size()) return false;
    unsigned long long acc = 0;
    for (; i < s.size(); i++) {
        int d;
        char c = s[i];
        if (c >= '0' && c <= '9') d = c - '0';
        else if (c >= 'a' && c <= 'z') d = c - 'a' + 10;
        else if (c >= 'A' && c <= 'Z') d = c - 'A' + 10;
        else return false;
        if (d >= base) return false;
        if (acc > (0x8000000000000000ull - d) / base + 1) return false;
        acc = acc * base + d;
        if (acc > 0x8000000000000000ull) return false;
    }
    if (!negv && acc > 0x7fffffffffffffffull) return false;
    out = negv ? (long long)(0 - acc) : (long long)acc;
    return true;
}
bool ieq(SV a, const char* b) {
    size_t n = std::strlen(b);
    if (a.size() != n) return false;
    for (size_t i = 0; i < n; i++) if (std::tolower((unsigned char)a[i]) != b[i]) return false;
    return true;
}
// Rust's <f64 as FromStr>: [+-]? (digits [. digits*]? | . digits) ([eE][+-]?digits)? | inf | infinity | nan
bool rust_parse_f64(SV s, double& out) {
    if (s.empty()) return false;
    size_t i = 0;
    if (s[0] == '+' || s[0] == '-') i = 1;
    SV body = s.substr(i);
    if (ieq(body, "inf") || ieq(body, "infinity")) { out = s[0] == '-' ? -std::numeric_limits<double>::infinity() : std::numeric_limits<double>::infinity(); return true; }
    if (ieq(body, "nan")) { out = std::numeric_limits<double>::quiet_NaN(); return true; }
    size_t j = i, nd = 0;
    while (j < s.size() && std::isdigit((unsigned char)s[j])) { j++; nd++; }
    if (j < s.size() && s[j] == '.') { j++; while (j < s.size() && std::isdigit((unsigned char)s[j])) { j++; nd++; } }
    if (nd == 0) return false;
    if (j < s.size() && (s[j] == 'e' || s[j] == 'E')) {
        j++;
        if (j < s.size() && (s[j] == '+' || s[j] == '-')) j++;
        size_t ne = 0;
        while (j < s.size() && std::isdigit((unsigned char)s[j])) { j++; ne++; }
        if (ne == 0) return false;
    }
    if (j != s.size()) return false;
    char buf[64];                                // strtod wants a terminator; correctly rounded, like Rust's parser
    if (s.size() < sizeof buf) { std::memcpy(buf, s.data(), s.size()); buf[s.size()] = 0; out = std::strtod(buf, nullptr); }
    else out = std::strtod(std::string(s).c_str(), nullptr);
    return true;
}
// yaml-rust parse_f64: .inf/.nan spellings, else Rust f64 parse
bool yaml_parse_f64(SV v, double& out) {
    if (v == ".inf" || v == ".Inf" || v == ".INF" || v == "+.inf" || v == "+.Inf" || v == "+.INF") { out = std::numeric_limits<double>::infinity(); return true; }
    if (v == "-.inf" || v == "-.Inf" || v == "-.INF") { out = -std::numeric_limits<double>::infinity(); return true; }
    if (v == ".nan" || v == "NaN" || v == ".NAN") { out = std::numeric_limits<double>::quiet_NaN(); return true; }
    return rust_parse_f64(v, out);
}
enum STy { TInt, TReal, TNull, TBool, TStr };
STy classify_uncached(const Y& y) {
    if (y.quoted) return TStr;
    const SV v = y.s;
    if (v.compare(0, 2, "0x") == 0 && parse_i64(v.substr(2), 16, y.iv)) return TInt;
    if (v.compare(0, 2, "0o") == 0 && parse_i64(v.substr(2), 8, y.iv)) return TInt;
    if (!v.empty() && v[0] == '+' && parse_i64(v.substr(1), 10, y.iv)) return TInt;
    if (v == "~" || v == "null") return TNull;
    if (v == "true" || v == "false") return TBool;
    if (parse_i64(v, 10, y.iv)) return TInt;
    if (yaml_parse_f64(v, y.dv)) return TReal;
    return TStr;
}
STy classify(const Y& y, long long* iv = nullptr, double* dv = nullptr) {
    if (y.ty < 0) y.ty = (signed char)classify_uncached(y);     // keys are typed once, not on every lookup
    if (iv) *iv = y.iv;
    if (dv) *dv = y.dv;
    return (STy)y.ty;
}

// ------------------------------------------------------------------ yaml.rs
struct Loader {
    World& world;
    std::vector<rtxh_camera>& cameras;
    std::map<std::string, rtxh_material, std::less<>> materials;   // MaterialStore (yaml.rs:344-346)
    std::map<std::string, int, std::less<>> groups;              // GroupStore (yaml.rs:281-283)

    [[noreturn]] void fail(const std::string& m) { throw ParseError{m}; }

    const Y* get(const Y* map, const char* key) {
        if (!map || map->t != Y::Map) return nullptr;
        for (auto& kv : map->kv)
            if (kv.first->t == Y::Scalar && kv.first->s == key && classify(*kv.first) == TStr) return kv.second;
        return nullptr;
    }
    SV as_str(const Y* y, const char* what) {                   // .as_str().unwrap()
        if (!y || y->t != Y::Scalar || classify(*y) != TStr) fail(std::string("expected a string for ") + what);
        return y->s;
    }
    bool opt_str(const Y* y, SV& out) {                          // .as_str()
        if (!y || y->t != Y::Scalar || classify(*y) != TStr) return false;
        out = y->s; return true;
    }
    SV key_str(const Y* k) {                                     // key.as_str().unwrap()
        if (k->t != Y::Scalar || classify(*k) != TStr) fail("mapping key is not a string");
        return k->s;
    }
    const Y& as_hash(const Y* y, const char* what) {
        if (!y || y->t != Y::Map) fail(std::string("expected a mapping for ") + what);
        return *y;
    }

    double f64_const(SV w) {                                     // yaml.rs:497-509
        double d;
        if (rust_parse_f64(w, d)) return d;
        if (w == "pi") return 3.14159265358979323846264338327950288;
        if (w == "-pi") return -3.14159265358979323846264338327950288;
        std::printf("Cannot parse as f64: \"%.*s\"\n", (int)w.size(), w.data());
        return std::numeric_limits<double>::quiet_NaN();
    }
    double f64_expr(const Y* y, const char* what) {               // yaml.rs:466-495
        if (!y || y->t != Y::Scalar) fail(std::string("expected a number for ") + what);
        long long iv; double dv;
        STy t = classify(*y, &iv, &dv);
        if (t == TReal) return dv;
        if (t == TInt) return (double)iv;
        if (t != TStr) fail(std::string("expected a number for ") + what);
        std::vector<SV> words;
        { size_t i = 0; const SV s = y->s;
          while (i < s.size()) { while (i < s.size() && std::isspace((unsigned char)s[i])) i++; size_t b = i;
              while (i < s.size() && !std::isspace((unsigned char)s[i])) i++; if (i > b) words.push_back(s.substr(b, i - b)); } }
        if (words.empty()) fail(std::string("empty expression for ") + what);
        double res = f64_const(words[0]);
        bool have_op = false; SV op;
        for (size_t i = 1; i < words.size(); i++) {
            if (!have_op) { have_op = true; op = words[i]; }      // `op` is never cleared (yaml.rs:474-491)
            else {
                double right = f64_const(words[i]);
                if (op == "+") res += right; else if (op == "-") res -= right;
                else if (op == "*") res *= right; else if (op == "/") res /= right;
            }
        }
        return res;
    }
    void f64_tuple(const Y* y, const char* what, double* out) {   // yaml.rs:236-244
        if (!y || y->t != Y::Seq) fail(std::string("expected a 3-element list for ") + what);
        if (y->items.size() != 3) fail(std::string("expected exactly 3 elements for ") + what);
        for (int i = 0; i < 3; i++) out[i] = f64_expr(y->items[i], what);
    }
    Mat4 transform(const Y* y) {                                   // yaml.rs:246-279
        if (!y || y->t != Y::Seq) fail("transform must be a list");
        Mat4 t = mat_identity();
        for (const Y* it : y->items) {
            const Y& h = as_hash(it, "transform entry");
            if (h.kv.size() != 1) fail("transform entry must have exactly one key");
            const SV key = key_str(h.kv[0].first);
            const Y* v = h.kv[0].second;
            double a[3];
            if (key == "scale") { f64_tuple(v, "scale", a); t = mat_mul(mat_scaling(a[0], a[1], a[2]), t); }
            else if (key == "translate") { f64_tuple(v, "translate", a); t = mat_mul(mat_translation(a[0], a[1], a[2]), t); }
            else if (key == "rotate-x") t = mat_mul(mat_rotation(0, f64_expr(v, "rotate-x")), t);
            else if (key == "rotate-y") t = mat_mul(mat_rotation(1, f64_expr(v, "rotate-y")), t);
            else if (key == "rotate-z") t = mat_mul(mat_rotation(2, f64_expr(v, "rotate-z")), t);
            else std::printf("unknown transform: %.*s\n", (int)key.size(), key.data());
        }
        return t;
    }
    void camera(const Y* e) {                                     // yaml.rs:75-87
        const Y* w = get(e, "width"); const Y* h = get(e, "height");
        long long wi, hi;
        if (!w || w->t != Y::Scalar || classify(*w, &wi) != TInt) fail("camera width must be an integer");
        if (!h || h->t != Y::Scalar || classify(*h, &hi) != TInt) fail("camera height must be an integer");
        const double fov = f64_expr(get(e, "field-of-view"), "field-of-view");
        rtxh_camera c; camera_new((int)wi, (int)hi, fov, &c);
        if (get(e, "up")) {
            double from[3], to[3], up[3];
            f64_tuple(get(e, "from"), "from", from); f64_tuple(get(e, "to"), "to", to); f64_tuple(get(e, "up"), "up", up);
            Mat4 v = mat_view(from, to, up); std::memcpy(c.transform, v.e, sizeof v.e);
        }
        cameras.push_back(c);
    }
    void pattern(const Y* e, rtxh_material& m) {                  // yaml.rs:427-464
        const SV ty = as_str(get(e, "type"), "pattern type");
        Mat4 id = mat_identity(); std::memcpy(m.pattern_transform, id.e, sizeof id.e);
        for (int i = 0; i < 3; i++) { m.color_a[i] = 0.0; m.color_b[i] = 0.0; }
        if (ty == "plain" || ty == "uniform") { m.pattern_kind = RTX_PAT_PLAIN; f64_tuple(get(e, "color"), "color", m.color_a); }
        else if (ty == "stripes" || ty == "rings" || ty == "checkers" || ty == "gradient") {
            m.pattern_kind = ty == "stripes" ? RTX_PAT_STRIPES : ty == "rings" ? RTX_PAT_RINGS
                           : ty == "checkers" ? RTX_PAT_CHECKERS : RTX_PAT_GRADIENT;
            f64_tuple(get(e, "color1"), "color1", m.color_a); f64_tuple(get(e, "color2"), "color2", m.color_b);
        } else {
            std::printf("unknown pattern type: %.*s\n", (int)ty.size(), ty.data());
            m.pattern_kind = RTX_PAT_PLAIN;                        // Pattern::uniform(black)
        }
        if (get(e, "transform")) { Mat4 t = transform(get(e, "transform")); std::memcpy(m.pattern_transform, t.e, sizeof t.e); }
    }
    void material(const Y* e) {                                   // yaml.rs:357-416
        rtxh_material m; default_material(&m);
        bool named = false; SV name;
        for (auto& kv : e->kv) {
            const SV key = key_str(kv.first);
            const Y* v = kv.second;
            if (key == "name") named = opt_str(v, name);
            else if (key == "extends") {
                auto it = materials.find(as_str(v, "extends"));
                if (it == materials.end()) fail("material extends unknown material '" + std::string(v->s) + "'");
                m = it->second;
            } else if (key == "color") {
                double c[3]; f64_tuple(v, "color", c);
                m.pattern_kind = RTX_PAT_PLAIN; Mat4 id = mat_identity(); std::memcpy(m.pattern_transform, id.e, sizeof id.e);
                for (int i = 0; i < 3; i++) { m.color_a[i] = c[i]; m.color_b[i] = 0.0; }
            } else if (key == "pattern") { as_hash(kv.second, "pattern"); pattern(kv.second, m); }
            else if (key == "ambient") m.ambient = f64_expr(v, "ambient");
            else if (key == "diffuse") m.diffuse = f64_expr(v, "diffuse");
            else if (key == "specular") m.specular = f64_expr(v, "specular");
            else if (key == "shininess") m.shininess = f64_expr(v, "shininess");
            else if (key == "reflective") m.reflective = f64_expr(v, "reflective");
            else if (key == "transparency") m.transparency = f64_expr(v, "transparency");
            else if (key == "refractive-index") m.refractive_index = f64_expr(v, "refractive-index");
            else if (key == "define") {}
            else std::printf("unknown material parameter: %.*s\n", (int)key.size(), key.data());
        }
        if (named) materials[std::string(name)] = m;
    }
    void group(const Y* e) {                                      // yaml.rs:294-333
        bool named = false; SV name; int parent = -1; Mat4 t = mat_identity();
        for (auto& kv : e->kv) {
            const SV key = key_str(kv.first);
            const Y* v = kv.second;
            if (key == "name") named = opt_str(v, name);
            else if (key == "parent") {
                auto it = groups.find(as_str(v, "parent"));
                if (it == groups.end()) fail("group parent '" + std::string(v->s) + "' is not defined");
                parent = it->second;
            } else if (key == "transform") t = transform(v);
            else if (key == "define") {}
            else std::printf("unknown group parameter: %.*s\n", (int)key.size(), key.data());
        }
        const int g = world.add_group(t);
        if (parent >= 0) world.set_group_of(OBJ_GROUP, g, parent); else world.add_object_to_scene(OBJ_GROUP, g);
        if (!named) fail("Group without name.");
        groups[std::string(name)] = g;
    }
    // yaml.rs:108-137; returns the ShapeIndex, sets group_name if the entry names a group
    int shape_properties(int type, const Y* e, bool& has_group, SV& group_name) {
        rtxh_material m; default_material(&m);
        Mat4 t = mat_identity();
        has_group = false;
        for (auto& kv : e->kv) {
            const SV key = key_str(kv.first);
            const Y* v = kv.second;
            if (key == "type" || key == "add") {}
            else if (key == "material") {
                auto it = materials.find(as_str(v, "material"));
                if (it == materials.end()) fail("unknown material '" + std::string(v->s) + "'");
                m = it->second;
            } else if (key == "transform") t = transform(v);
            else if (key == "group") has_group = opt_str(v, group_name);
            else std::printf("unknown shape parameter: %.*s\n", (int)key.size(), key.data());
        }
        return world.add_shape(type, &m, t);
    }
    void toplevel_shape(int type, const Y* e) {                   // yaml.rs:95-106
        bool hg; SV gn;
        const int s = shape_properties(type, e, hg, gn);
        if (hg) {
            auto it = groups.find(gn);
            if (it == groups.end()) fail("shape refers to unknown group '" + std::string(gn) + "'");
            world.set_group_of(OBJ_SHAPE, s, it->second);
        } else world.add_object_to_scene(OBJ_SHAPE, s);
    }
    void object(const Y* e, int& kind, int& index) {              // yaml.rs:139-171
        as_hash(e, "CSG operand");
        if (!get(e, "add")) fail("Expected keyword 'add' not found.");
        const SV what = as_str(get(e, "add"), "add");
        bool hg = false; SV gn;
        if (what == "plane") { kind = OBJ_SHAPE; index = shape_properties(RTX_NODE_PLANE, e, hg, gn); }
        else if (what == "sphere") { kind = OBJ_SHAPE; index = shape_properties(RTX_NODE_SPHERE, e, hg, gn); }
        else if (what == "cube") { kind = OBJ_SHAPE; index = shape_properties(RTX_NODE_CUBE, e, hg, gn); }
        else if (what == "csg") { kind = OBJ_CSG; index = csg(e); }
        else if (what == "group") fail("not implemented: group as CSG operand");
        else fail("trying to add unknown object: " + std::string(what));
        if (hg) fail("CSG operands may not name a group");
    }
    int csg(const Y* e) {                                         // yaml.rs:173-216
        Mat4 t = mat_identity();
        int op = -1; bool have_operands = false; int lk = 0, li = 0, rk = 0, ri = 0;
        for (auto& kv : e->kv) {
            const SV key = key_str(kv.first);
            const Y* v = kv.second;
            if (key == "operator") {
                const SV o = as_str(v, "operator");
                if (o == "union") op = RTX_CSG_UNION; else if (o == "intersection") op = RTX_CSG_INTERSECTION;
                else if (o == "difference") op = RTX_CSG_DIFFERENCE; else fail("Unknown CSG operator: " + std::string(o));
            } else if (key == "transform") t = transform(v);
            else if (key == "operands") {
                if (v->t != Y::Seq) fail("operands must be a list");
                if (v->items.size() != 2) fail("CSG needs exactly two operands");
                object(v->items[0], lk, li); object(v->items[1], rk, ri);
                have_operands = true;
            } else if (key == "add") {}
            else std::printf("unknown csg parameter: %.*s\n", (int)key.size(), key.data());
        }
        if (op < 0) fail("Missing operator of CSG");
        if (!have_operands) fail("Missing operands of CSG");
        return world.add_csg(op, lk, li, rk, ri, t);
    }
    void document(const Y* doc) {                                 // yaml.rs:22-73
        if (!doc || doc->t != Y::Seq) fail("scene file must be a YAML sequence");
        world.nodes.reserve(world.nodes.size() + doc->items.size());
        for (const Y* e : doc->items) {
            as_hash(e, "scene entry");
            if (get(e, "add")) {
                const SV what = as_str(get(e, "add"), "add");
                if (what == "camera") camera(e);
                else if (what == "light") {                        // yaml.rs:89-93
                    double at[3], in[3]; f64_tuple(get(e, "at"), "at", at); f64_tuple(get(e, "intensity"), "intensity", in);
                    world.add_light(at, in);
                } else if (what == "plane") toplevel_shape(RTX_NODE_PLANE, e);
                else if (what == "sphere") toplevel_shape(RTX_NODE_SPHERE, e);
                else if (what == "cube") toplevel_shape(RTX_NODE_CUBE, e);
                else if (what == "csg") { const int c = csg(e); world.add_object_to_scene(OBJ_CSG, c); }
                else std::printf("trying to add unknown object: %.*s\n", (int)what.size(), what.data());
            } else if (get(e, "define")) {
                const SV what = as_str(get(e, "define"), "define");
                if (what == "material") material(e);
                else if (what == "group") group(e);
                else std::printf("trying to define unknown property: %.*s\n", (int)what.size(), what.data());
            }
        }
        world.build();
    }
};

}  // namespace

bool load_yaml(const std::string& text, World& world, std::vector<rtxh_camera>& cameras, std::string& err) {
    try {
        Reader r(text);
        YP doc = r.document();
        Loader l{world, cameras, {}, {}};
        l.document(doc);
        return true;
    } catch (const ParseError& e) {
        err = e.msg;
        return false;
    }
}

}  // namespace rth
