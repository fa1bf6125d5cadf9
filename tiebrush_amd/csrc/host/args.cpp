#include "args.h"

#include <stdio.h>
#include <string.h>

Args::Args(int argc, char** argv, const char* fmt) {
  std::map<std::string, bool> takes;  // option -> needs a value
  std::string f(fmt);
  size_t s = 0;
  std::string shorts;
  for (;;) {
    size_t e = f.find(';', s);
    if (e == std::string::npos) {
      shorts = f.substr(s);
      break;
    }
    std::string w = f.substr(s, e - s);
    bool val = !w.empty() && w.back() == '=';
    if (val) w.pop_back();
    takes[w] = val;
    s = e + 1;
  }
  for (size_t i = 0; i < shorts.size(); ++i) {
    bool val = i + 1 < shorts.size() && shorts[i + 1] == ':';
    takes[std::string(1, shorts[i])] = val;
    if (val) ++i;
  }
  for (int i = 0; i < argc; ++i) argv_.push_back(argv[i]);
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
      std::string name = a.substr(2), val;
      size_t eq = name.find('=');
      bool has = eq != std::string::npos;
      if (has) {
        val = name.substr(eq + 1);
        name.resize(eq);
      }
      auto it = takes.find(name);
      if (it == takes.end() || name.size() < 2) {
        err_ = "Error: invalid argument: " + a;
        return;
      }
      if (it->second && !has) {
        if (i + 1 >= argc) {
          err_ = "Error: value required for option --" + name;
          return;
        }
        val = argv[++i];
      }
      opts_[name] = val;
    } else if (a.size() > 1 && a[0] == '-' && a != "-") {
      for (size_t k = 1; k < a.size(); ++k) {
        std::string name(1, a[k]);
        auto it = takes.find(name);
        if (it == takes.end()) {
          err_ = "Error: invalid argument: " + a;
          return;
        }
        if (it->second) {
          std::string val = a.substr(k + 1);
          if (!val.empty() && val[0] == '=') val = val.substr(1);
          if (val.empty()) {
            if (i + 1 >= argc) {
              err_ = "Error: value required for option -" + name;
              return;
            }
            val = argv[++i];
          }
          opts_[name] = val;
          break;
        }
        opts_[name] = "";
      }
    } else {
      nonopt_.push_back(a);
    }
  }
}

const char* Args::getOpt(const char* name) const {
  auto it = opts_.find(name);
  return it == opts_.end() ? nullptr : it->second.c_str();
}

void Args::printCmdLine(FILE* f) const {
  for (size_t i = 0; i < argv_.size(); ++i) fprintf(f, "%s%s", argv_[i].c_str(), i + 1 < argv_.size() ? " " : "\n");
}
