/* factories.h - name -> constructor registries (models, noise models, inference techniques,
 * convergence detectors). Same public surface as the reference's TemplateFactory /
 * SingletonFactory / FactoryRegistration (factories.h:19-174) so that model sources using
 * `FactoryRegistration<FwdModelFactory, MyModel> registration("name")` link unchanged. */
#pragma once

#include <map>
#include <string>
#include <vector>

template <class T> class TemplateFactory
{
public:
    typedef T *(*Function)(void);
    void Add(const std::string &name, Function function)
    {
        m_makers[name] = function;
    }
    T *Create(const std::string &name)
    {
        typename std::map<std::string, Function>::iterator it = m_makers.find(name);
        return (it == m_makers.end()) ? NULL : it->second();
    }
    std::vector<std::string> GetNames()
    {
        std::vector<std::string> names;
        for (typename std::map<std::string, Function>::iterator it = m_makers.begin(); it != m_makers.end(); ++it)
            names.push_back(it->first);
        return names;
    }
    bool HasName(const std::string &name)
    {
        return m_makers.count(name) > 0;
    }

private:
    std::map<std::string, Function> m_makers;
};

template <class T> class SingletonFactory : public TemplateFactory<T>
{
public:
    static SingletonFactory *GetInstance()
    {
        if (!s_instance)
            s_instance = new SingletonFactory<T>();
        return s_instance;
    }
    static void Destroy()
    {
        delete s_instance;
        s_instance = NULL;
    }

private:
    SingletonFactory()
    {
    }
    static SingletonFactory *s_instance;
};
template <class T> SingletonFactory<T> *SingletonFactory<T>::s_instance = NULL;

/** Static-initialisation helper: `static FactoryRegistration<Factory, Class> reg("name");` */
template <class T, class U> class FactoryRegistration
{
public:
    explicit FactoryRegistration(std::string name)
    {
        (T::GetInstance())->Add(name, &U::NewInstance);
    }
};
