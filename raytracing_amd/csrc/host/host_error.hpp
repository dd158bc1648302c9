// last-error text of the host layer (returned by rt_host_last_error), one per thread
#pragma once
#include <string>
namespace rt {
void setHostError(const std::string &msg);
}
