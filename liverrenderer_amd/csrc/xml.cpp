#include "xml.h"
#include <cctype>

namespace lrt {

namespace {
struct Parser {
    const std::string &s; size_t i = 0;
    explicit Parser(const std::string &t) : s(t) {}
    [[noreturn]] void fail(const std::string &msg) {
        size_t line = 1; for (size_t k = 0; k < i && k < s.size(); ++k) if (s[k] == '\n') ++line;
        throw std::runtime_error("XML parse error (line " + std::to_string(line) + "): " + msg);
    }
    void skip_ws() { while (i < s.size() && isspace((unsigned char) s[i])) ++i; }
    bool starts(const char *p) const { return s.compare(i, strlen_(p), p) == 0; }
    static size_t strlen_(const char *p) { size_t n = 0; while (p[n]) ++n; return n; }
    void skip_misc() {                      // whitespace, comments, prolog, text
        for (;;) {
            while (i < s.size() && s[i] != '<') ++i;
            if (i >= s.size()) return;
            if (starts("<!--")) { size_t e = s.find("-->", i + 4); if (e == std::string::npos) fail("unterminated comment"); i = e + 3; continue; }
            if (starts("<?")) { size_t e = s.find("?>", i + 2); if (e == std::string::npos) fail("unterminated prolog"); i = e + 2; continue; }
            if (starts("<!")) { size_t e = s.find('>', i); if (e == std::string::npos) fail("unterminated declaration"); i = e + 1; continue; }
            return;
        }
    }
    std::string name() {
        size_t b = i;
        while (i < s.size() && (isalnum((unsigned char) s[i]) || s[i] == '_' || s[i] == '-' || s[i] == ':' || s[i] == '.')) ++i;
        if (b == i) fail("expected a name");
        return s.substr(b, i - b);
    }
    static std::string unescape(const std::string &v) {
        std::string o; o.reserve(v.size());
        for (size_t k = 0; k < v.size(); ++k) {
            if (v[k] == '&') {
                if (!v.compare(k, 4, "&lt;")) { o += '<'; k += 3; continue; }
                if (!v.compare(k, 4, "&gt;")) { o += '>'; k += 3; continue; }
                if (!v.compare(k, 5, "&amp;")) { o += '&'; k += 4; continue; }
                if (!v.compare(k, 6, "&quot;")) { o += '"'; k += 5; continue; }
                if (!v.compare(k, 6, "&apos;")) { o += '\''; k += 5; continue; }
            }
            o += v[k];
        }
        return o;
    }
    std::unique_ptr<XmlNode> element() {
        if (s[i] != '<') fail("expected '<'");
        ++i;
        auto n = std::make_unique<XmlNode>();
        n->tag = name();
        for (;;) {
            skip_ws();
            if (i >= s.size()) fail("unexpected end of file in tag");
            if (s[i] == '/') { if (i + 1 >= s.size() || s[i + 1] != '>') fail("malformed empty-element tag"); i += 2; return n; }
            if (s[i] == '>') { ++i; break; }
            std::string k = name();
            skip_ws(); if (i >= s.size() || s[i] != '=') fail("expected '='"); ++i; skip_ws();
            if (i >= s.size() || (s[i] != '"' && s[i] != '\'')) fail("expected quoted attribute value");
            char q = s[i++]; size_t e = s.find(q, i); if (e == std::string::npos) fail("unterminated attribute value");
            n->attrs.emplace_back(k, unescape(s.substr(i, e - i))); i = e + 1;
        }
        for (;;) {
            skip_misc();
            if (i >= s.size()) fail("missing closing tag for <" + n->tag + ">");
            if (starts("</")) {
                i += 2; std::string c = name(); skip_ws();
                if (c != n->tag) fail("mismatched closing tag </" + c + "> for <" + n->tag + ">");
                if (i >= s.size() || s[i] != '>') fail("expected '>'");
                ++i; return n;
            }
            n->children.push_back(element());
        }
    }
};
} // namespace

std::unique_ptr<XmlNode> xml_parse(const std::string &text) {
    Parser p(text);
    p.skip_misc();
    if (p.i >= text.size()) p.fail("no root element");
    return p.element();
}

} // namespace lrt
