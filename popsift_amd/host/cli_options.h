/*
 * cli_options.h -- the small command-line parser of the demo programs.
 * The reference uses boost::program_options (main.cpp:48-146, match.cpp:47-146); Boost is not a
 * dependency of this build.  Same surface: `--name value`, `--name=value`, `-x value`, bool switches
 * without a value, unambiguous prefixes of long names (program_options' default "allow_guessing"),
 * grouped --help text, "required option missing" and "unrecognised option" errors.
 */
#pragma once

#include <cstdlib>
#include <functional>
#include <iostream>
#include <string>
#include <vector>

namespace cli {

struct Option {
    std::string                             name;  /* long name without the dashes */
    char                                    shrt;  /* 0 = none */
    bool                                    takes_value;
    bool                                    required;
    std::string                             group;
    std::string                             help;
    std::function<void(const std::string&)> apply; /* value, or "" for a switch */
};

class Options {
public:
    void flag(const std::string& n, char s, const std::string& g, const std::string& h, std::function<void()> f)
    {
        _opts.push_back(Option{n, s, false, false, g, h, [f](const std::string&) { f(); }});
    }
    void val(const std::string& n, char s, const std::string& g, const std::string& h,
             std::function<void(const std::string&)> f, bool required = false)
    {
        _opts.push_back(Option{n, s, true, required, g, h, f});
    }
    void fval(const std::string& n, const std::string& g, const std::string& h, std::function<void(float)> f)
    {
        val(n, 0, g, h, [this, n, f](const std::string& v) {
            char*       end = 0;
            const float x = strtof(v.c_str(), &end);
            if (end == v.c_str() || *end != 0) error("the argument ('" + v + "') for option '--" + n + "' is invalid");
            f(x);
        });
    }
    void ival(const std::string& n, const std::string& g, const std::string& h, std::function<void(int)> f)
    {
        val(n, 0, g, h, [this, n, f](const std::string& v) {
            char*      end = 0;
            const long x = strtol(v.c_str(), &end, 10);
            if (end == v.c_str() || *end != 0) error("the argument ('" + v + "') for option '--" + n + "' is invalid");
            f((int)x);
        });
    }

    void print(std::ostream& o) const
    {
        std::string group;
        for (const Option& op : _opts) {
            if (op.group != group) {
                group = op.group;
                o << std::endl << group << ":" << std::endl;
            }
            std::string left = "  ";
            if (op.shrt) left += std::string("-") + op.shrt + " [ --" + op.name + " ]";
            else left += "--" + op.name;
            if (op.takes_value) left += " arg";
            o << left;
            if (!op.help.empty()) {
                if (left.size() < 28) o << std::string(28 - left.size(), ' ');
                else o << std::endl << std::string(28, ' ');
                for (char c : op.help) { /* continuation lines are indented like the first */
                    o << c;
                    if (c == '\n') o << std::string(28, ' ');
                }
            }
            o << std::endl;
        }
    }

    /* main.cpp:139-144 */
    [[noreturn]] void error(const std::string& what) const
    {
        std::cerr << "Error: " << what << std::endl << std::endl << "Usage:\n\nAllowed options:";
        print(std::cerr);
        std::cerr << std::endl;
        exit(EXIT_FAILURE);
    }

    /* parses; `--help` prints the table on stdout and exits with 1 (main.cpp:132-135) */
    void parse(int argc, char** argv)
    {
        bool                     want_help = false;
        std::vector<std::string> seen;
        for (int i = 1; i < argc; i++) {
            const std::string a = argv[i];
            const Option*     op = 0;
            std::string       value;
            bool              has_value = false;
            if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
                std::string  name = a.substr(2);
                const size_t eq = name.find('=');
                if (eq != std::string::npos) {
                    value = name.substr(eq + 1);
                    name = name.substr(0, eq);
                    has_value = true;
                }
                op = find_long(name);
                if (!op) error("unrecognised option '--" + name + "'");
            } else if (a.size() >= 2 && a[0] == '-' && a[1] != '-') {
                for (const Option& o : _opts)
                    if (o.shrt == a[1]) op = &o;
                if (!op) error("unrecognised option '" + a + "'");
                if (a.size() > 2) { /* -ifile */
                    value = a.substr(2);
                    has_value = true;
                }
            } else {
                error("too many positional options have been specified on the command line");
            }
            if (op->name == "help") {
                want_help = true;
                continue;
            }
            if (op->takes_value) {
                if (!has_value) {
                    if (i + 1 >= argc) error("the required argument for option '--" + op->name + "' is missing");
                    value = argv[++i];
                }
                seen.push_back(op->name);
                op->apply(value);
            } else {
                if (has_value) error("option '--" + op->name + "' does not take any arguments");
                op->apply("");
            }
        }
        if (want_help) {
            std::cout << "Allowed options:";
            print(std::cout);
            std::cout << '\n';
            exit(1);
        }
        for (const Option& op : _opts) {
            if (!op.required) continue;
            bool ok = false;
            for (const std::string& s : seen) ok = ok || s == op.name;
            if (!ok) error("the option '--" + op.name + "' is required but missing");
        }
    }

private:
    const Option* find_long(const std::string& name) const
    {
        const Option* hit = 0;
        int           hits = 0;
        for (const Option& op : _opts) {
            if (op.name == name) return &op;
            if (op.name.compare(0, name.size(), name) == 0) {
                hit = &op;
                hits++;
            }
        }
        if (hits > 1) error("option '--" + name + "' is ambiguous");
        return hits == 1 ? hit : 0;
    }

    std::vector<Option> _opts;
};

}  // namespace cli
