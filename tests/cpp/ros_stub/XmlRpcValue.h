// NOT ROS (see README.md)
#pragma once
#include <map>
#include <string>
namespace XmlRpc {
class XmlRpcValue {
public:
    enum Type { TypeInvalid, TypeBoolean, TypeInt, TypeDouble, TypeString, TypeDateTime, TypeBase64, TypeArray, TypeStruct };
    typedef std::map<std::string, XmlRpcValue> ValueStruct;
    typedef ValueStruct::iterator iterator;
    Type getType() const;
    int size() const;
    operator bool&(); operator int&(); operator double&(); operator std::string&();
    XmlRpcValue const& operator[](int i) const;
    XmlRpcValue& operator[](int i);
    iterator begin(); iterator end();
};
}  // namespace XmlRpc
