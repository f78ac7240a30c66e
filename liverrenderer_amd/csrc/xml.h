// Minimal XML reader for the Mitsuba scene-format subset the liver scenes use
// (elements, attributes, comments, <?xml?> prolog; no DTD, no CDATA).
// Replaces pugixml as used by src/core/parser.cpp in the reference.
#pragma once
#include <string>
#include <vector>
#include <map>
#include <memory>
#include <stdexcept>

namespace lrt {

struct XmlNode {
    std::string tag;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<XmlNode>> children;

    const std::string *find(const std::string &k) const {
        for (auto &a : attrs) if (a.first == k) return &a.second;
        return nullptr;
    }
    std::string get(const std::string &k, const std::string &def = "") const {
        const std::string *v = find(k); return v ? *v : def;
    }
    bool has(const std::string &k) const { return find(k) != nullptr; }
};

std::unique_ptr<XmlNode> xml_parse(const std::string &text);   // throws std::runtime_error

} // namespace lrt
